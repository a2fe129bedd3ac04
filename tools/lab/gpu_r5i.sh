#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -q -x 2>&1 | tail -3
for rep in 1 2; do for fl in 0 33554432; do
  timeout 300 python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --debug-flags $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('flags $fl: %.3f ms'%d['ms_per_step'], {n:(round(v['launches_per_step']),round(v['avg_launch_us'],1)) for n,v in k.items() if n.startswith(('pw_wgrad'))})"
done; done
timeout 300 python bench.py --train --workload cfg4_improved_u36_n2048 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('cfg4: %.3f ms'%d['ms_per_step'], {n:(round(v['launches_per_step']),round(v['avg_launch_us'],1)) for n,v in k.items() if n.startswith(('pw_wgrad'))})"
