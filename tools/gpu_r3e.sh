#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03e}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export GEMM_ROUNDS=5 GEMM_ITERS=10
c() { echo $(( ($1 + 1) << 26 )); }
GEMM_SHAPES=proj_1x1,bottleneck,cfg4_proj,cfg4_bottleneck timeout 900 python tools/gemm_ab.py cp0=$(c 0) cp1=$(c 1) cp4=$(c 4) cp5=$(c 5) > "$OUT/cp_a.log" 2>&1
grep -v "^{" "$OUT/cp_a.log" | tail -30
GEMM_SHAPES=res_conv,mask,cfg4_res_conv,cfg5_res_conv,cfg5_mask timeout 900 python tools/gemm_ab.py cp0=$(c 0) cp1=$(c 1) cp4=$(c 4) cp5=$(c 5) cp8=$(c 8) cp12=$(c 12) cp13=$(c 13) > "$OUT/cp_b.log" 2>&1
grep -v "^{" "$OUT/cp_b.log" | tail -40
echo "== done"
