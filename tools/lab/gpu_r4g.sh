#!/bin/bash
# round 4: the paired-block GEMM (lab build) INSIDE the forward, with the two-stream split on: does a lighter GEMM block
# (80 KB of LDS, 128 registers) let the other stream's pyramid kernels co-reside on the CU?
set -u
OUT=gpurun_out/${1:-r04s}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LAB=$GRAFT_REPO_ROOT/sudo_rm_rf_amd/libsudormrf_hip_lab.so
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>$OUT/$name.err | tail -1 > $OUT/$name.json
  python - $OUT/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print("%-28s %.3f ms  %.0f sep-s/s  split %s" % (sys.argv[2], d["ms_per_step"], d["value"], d["config"].get("stream_split")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
  run shipped_$rep A=1
  run lab_x3w_$rep SRF_LIB=$LAB
  run lab_x3p512_$rep SRF_LIB=$LAB SRF_GEMM=x3p
  run lab_x3p256_$rep SRF_LIB=$LAB SRF_GEMM=x3p SRF_X3P_BLOCKS=256
  run lab_x3p384_$rep SRF_LIB=$LAB SRF_GEMM=x3p SRF_X3P_BLOCKS=384
  run lab_x3p256_half_$rep SRF_LIB=$LAB SRF_GEMM=x3p SRF_X3P_BLOCKS=256 SRF_STREAM_SPLIT=half
  run lab_x3p512_off_$rep SRF_LIB=$LAB SRF_GEMM=x3p SRF_STREAM_SPLIT=off
  run shipped_off_$rep SRF_STREAM_SPLIT=off
done
