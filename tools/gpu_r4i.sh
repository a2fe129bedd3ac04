#!/bin/bash
# round 4: which batch split suits the paired-block GEMM (product build)?  SRF_STREAM_SPLIT variants, cfg 2 (and cfg 4), same box
set -u
OUT=gpurun_out/${1:-r04v}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for w in cfg2_improved_u16 cfg4_improved_u36_n2048; do
for sp in auto off 1:1 5:3 3:1 1:1:1 2:1:1 1:1:1:1 3:3:2; do
  SRF_STREAM_SPLIT=$sp timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>$OUT/${w}_$sp.err | tail -1 > $OUT/${w}_$sp.json
  python - $OUT/${w}_$sp.json "$w $sp" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("%-40s %.3f ms  %.0f  split %s" % (sys.argv[2], d["ms_per_step"], d["value"], d["config"].get("stream_split")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done
