#!/bin/bash
# Round-2 diagnostics call 1: PRO-1 buffer-load GEMM error pattern, GroupComm two-stream bisect, counter list, SQ PMC pass.
set -u
OUT=gpurun_out/r02_diag1
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|Counter_Name)|SQ_|TCC_|GRBM_|TCP_|TA_" | head -400 ) > $OUT/counters.txt 2>&1
echo "== diag_pro1"; timeout 300 python tools/diag_pro1.py > $OUT/diag_pro1.log 2>&1; echo "rc=$?"; tail -30 $OUT/diag_pro1.log
echo "== diag_gc_split cfg3"; timeout 420 python tools/diag_gc_split.py cfg3_groupcomm_u8 30 5:3 > $OUT/diag_gc_cfg3.log 2>&1; echo "rc=$?"; cat $OUT/diag_gc_cfg3.log | tail -60
echo "== diag_gc_split cfg2 control"; ONLY=default timeout 200 python tools/diag_gc_split.py cfg2_improved_u16 30 5:3 > $OUT/diag_gc_cfg2.log 2>&1; echo "rc=$?"; tail -5 $OUT/diag_gc_cfg2.log
echo "== SQ pmc"
( cd /tmp && SRF_STREAM_SPLIT=off timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -o g -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile ) > $OUT/pmc_sq.log 2>&1
echo "pmc rc=$?"; tail -3 $OUT/pmc_sq.log
python - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + '/pmc_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'srf_' not in k: continue
        k = k.split('(')[0].replace('void ', '')
        e = agg[k][r['Counter_Name']]; e[0] += 1; e[1] += float(r['Counter_Value'])
with open(out + '/pmc_sq_summary.txt', 'w') as fo:
    for k, cs in agg.items():
        fo.write(k + "\n")
        for c, (n, v) in sorted(cs.items()):
            fo.write("   %-32s avg/launch %16.1f  (n=%d)\n" % (c, v / n, n))
print(open(out + '/pmc_sq_summary.txt').read()[:6000])
PY
find $OUT -name "*.csv" -size +20M -delete
echo "== done"
