#!/bin/bash
# PMC pass over bench.py (kernel filter given as $2 regex); counters groups as remaining args
OUT=gpurun_out/${1:-pmc}; FILT=$2; shift 2; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; i=1
for grp in "$@"; do
  rocprofv3 --kernel-trace --output-format csv --pmc ${grp//,/ } -d $GRAFT_REPO_ROOT/$OUT/p$i -o g -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1
  echo "pass $i rc=$?"; i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python - "$OUT" "$FILT" <<'PY'
import csv, sys, glob, collections, re
out, filt = sys.argv[1], sys.argv[2]
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(out+'/p*/g_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if not re.search(filt,k): continue
        k=k.split('(')[0].replace('void ','')
        e=agg[k][r['Counter_Name']]; e[0]+=1; e[1]+=float(r['Counter_Value'])
for k,cs in agg.items():
    print(k)
    for c,(n,v) in sorted(cs.items()):
        print("   %-28s avg/launch %14.1f  (n=%d)"%(c,v/n,n))
PY
find $OUT -name "*.csv" -size +20M -delete
