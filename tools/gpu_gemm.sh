#!/bin/bash
# GEMM iteration loop on the GPU box: unit parity of the pointwise convs, then the micro-benchmark (old persistent 128x128 vs packed 256x128)
set -u
OUT=gpurun_out/${1:-gemm}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --tb=short -x -k "pw_conv" --timeout 60 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log | cut -c1-250
GEMM_ITERS=${GEMM_ITERS:-30} timeout 200 python tools/gemm_bench.py ${MODES:-0u 0} 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tee $OUT/gemm_bench.log
