#!/bin/bash
# round 4, backward streaming kernels: parity of the changed kernels, then same-box A/B against the previous commit's build
set -u
OUT=gpurun_out/${1:-r04i}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py "tests/test_gpu_ops.py::test_pw_conv_three_part_split" -x -q -m gpu > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 200 python tools/f16_split_probe.py > $OUT/f16_split_probe.txt 2>&1; tail -9 $OUT/f16_split_probe.txt
timeout 1500 tools/train_lib_ab.sh ${1:-r04i} "cfg2_improved_u16 cfg4_improved_u36_n2048" base=$GRAFT_REPO_ROOT/tools/ab/libsudormrf_hip_base.so new=- new_oldmap=-:4096
