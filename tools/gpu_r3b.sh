#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03b}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export GEMM_SHAPES=proj_1x1 GEMM_ROUNDS=5 GEMM_ITERS=10
a() { echo $(( $1 << 16 )); }
timeout 900 python tools/gemm_ab.py full=0 noload=$(a 3) nomfma=$(a 4) nomfma_noload=$(a 7) nosplit_nomfma=$(a 12) noepi=$(a 16) noload_noepi=$(a 19) nomfma_noepi=$(a 20) nomfma_noload_noepi=$(a 23) all_but_frags=$(a 31) all_but_split=$(a 55) nothing=$(a 63) > "$OUT/ablate2.log" 2>&1
grep -v "^{" "$OUT/ablate2.log" | tail -30
export GEMM_SHAPES=res_conv
timeout 900 python tools/gemm_ab.py full=0 nomfma=$(a 4) nomfma_noepi=$(a 20) nomfma_noload_noepi=$(a 23) > "$OUT/ablate3.log" 2>&1
grep -v "^{" "$OUT/ablate3.log" | tail -30
echo "== done"
