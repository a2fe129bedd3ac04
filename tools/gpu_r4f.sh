#!/bin/bash
# round 4: the paired-block GEMM (csrc/experiments/srf_pwconv_x3p.hip, lab build) against the shipped kernel, bit for bit and in time
set -u
OUT=gpurun_out/${1:-r04k}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 SRF_LIB=$GRAFT_REPO_ROOT/sudo_rm_rf_amd/libsudormrf_hip_lab.so
GEMM_SHAPES=${GEMM_SHAPES:-proj_1x1,res_conv,bottleneck,proj_4096,res_conv_4096,cfg4_proj,cfg4_res_conv,cfg4_bottleneck} GEMM_ROUNDS=5 GEMM_ITERS=10 \
  timeout 600 python tools/gemm_ab.py ${VARIANTS:-x3w=0 x3p=0:0:x3p x3p_noepi=0:1:x3p x3p_nomfma=0:2:x3p} > $OUT/gemm_ab.log 2>&1
echo "rc=$?"; grep -v "^{" $OUT/gemm_ab.log | tail -${TAILN:-40}
