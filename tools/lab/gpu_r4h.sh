#!/bin/bash
# round 4: the paired-block GEMM as the product default -- full GPU tests, then same-box A/B of every configuration against
# debug flag 8192 (the one-block-per-CU kernel everywhere), forward and training step
set -u
OUT=gpurun_out/${1:-r04t}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -x -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
run() {  # name, args...
  local name=$1; shift
  timeout 400 python bench.py "$@" --no-cpu-baseline --no-kernel-profile 2>$OUT/$name.err | tail -1 > $OUT/$name.json
  python - $OUT/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("%-40s %.3f ms  %.0f  split %s" % (sys.argv[2], d["ms_per_step"], d["value"], d["config"].get("stream_split")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
  for w in cfg2_improved_u16 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096 cfg3_groupcomm_u8 cfg1_improved_u8; do
    run fwd_${w}_x3p_$rep --workload $w --steps 20 --warmup 5
    run fwd_${w}_x3w_$rep --workload $w --steps 20 --warmup 5 --debug-flags 8192
  done
  for w in cfg2_improved_u16 cfg4_improved_u36_n2048; do
    run train_${w}_x3p_$rep --train --workload $w --steps 8 --warmup 3
    run train_${w}_x3w_$rep --train --workload $w --steps 8 --warmup 3 --debug-flags 8192
  done
done
