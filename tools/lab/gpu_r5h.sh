#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -q -x 2>&1 | tail -4
for rep in 1 2; do for fl in 0 1; do
  timeout 300 python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --debug-flags $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('flags $fl: %.3f ms'%d['ms_per_step'], {n:(round(v['launches_per_step']),round(v['avg_launch_us'],1)) for n,v in k.items() if n.startswith(('pw_pair','pw_conv_x3w'))})"
done; done
