#!/bin/bash
# round 4: x3p epilogue through LDS strips (new library) against dword stores straight from the MFMA layout (tools/ab/..._direct.so)
set -u
OUT=gpurun_out/${1:-r04y}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
DIRECT=$GRAFT_REPO_ROOT/tools/ab/libsudormrf_hip_direct.so
for lib in "$DIRECT" ""; do
  echo "== ${lib:-strip (in-tree)}"
  SRF_LIB=$lib GEMM_SHAPES=proj_1x1,res_conv,bottleneck,cfg4_proj,cfg4_res_conv GEMM_ROUNDS=3 timeout 300 python tools/gemm_ab.py x3w=0 x3p=8192 2>&1 | grep -v "^{\|amdgpu.ids"
done
for rep in 1 2; do
  for lib in "$DIRECT" ""; do
    for w in cfg2_improved_u16 cfg4_improved_u36_n2048; do
      SRF_LIB=$lib timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-8s %-26s %.3f ms  %.0f  split %s' % ('${lib:+direct}' or 'strip', '$w', d['ms_per_step'], d['value'], d['config'].get('stream_split')))"
    done
  done
done
