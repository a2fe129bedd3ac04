#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "tac" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -k "groupcomm or cfg3 or gc" 2>&1 | tail -3
for rep in 1 2; do for fl in 0 1048576; do
  timeout 300 python bench.py --workload cfg3_groupcomm_u8 --steps 20 --warmup 6 --no-cpu-baseline --debug-flags $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('flags $fl: %.3f ms %s'%(d['ms_per_step'], d['config']['stream_split']), {n:round(v['avg_launch_us'],1) for n,v in k.items() if n.startswith(('tac','pw_conv_small'))}, 'ok' if d['self_check']['ok'] else 'BAD')"
done; done
