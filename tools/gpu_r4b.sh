#!/bin/bash
# Round 4: the role-split GEMM (srf_pwconv_x3s.hip) against round 3's, bit for bit and in time.
set -u
OUT=gpurun_out/${1:-r04b}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== gemm A/B: x3w vs x3s"
GEMM_SHAPES=${GEMM_SHAPES:-proj_1x1,res_conv,bottleneck,mask,cfg4_proj,cfg4_res_conv,cfg4_bottleneck} GEMM_ROUNDS=5 GEMM_ITERS=10 timeout 600 python tools/gemm_ab.py x3w=0 x3s=0:0:x3s > "$OUT/gemm_ab.log" 2>&1; echo "rc=$?"; grep -v "^{" "$OUT/gemm_ab.log" | tail -20
echo "== done"
