#!/bin/bash
# PMC profile of the GEMM micro-benchmark (counters in their own pass, kernel-trace only).
OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters_list.txt 2>&1
i=0
for grp in "$@"; do
  [ $i -eq 0 ] && { i=1; continue; }
  rocprofv3 --kernel-trace --output-format csv --pmc ${grp//,/ } -d $GRAFT_REPO_ROOT/$OUT/p$i -o g -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py 0 > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(out+'/p*/g_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'srf_pw' not in k: continue
        k=k.split('(')[0].replace('void ','')
        e=agg[k][r['Counter_Name']]; e[0]+=1; e[1]+=float(r['Counter_Value'])
for k,cs in agg.items():
    print(k)
    for c,(n,v) in sorted(cs.items()):
        print("   %-28s avg/launch %14.1f  (n=%d)"%(c,v/n,n))
PY
