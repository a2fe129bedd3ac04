#!/bin/bash
# bench.py cfg2 single-stream under several debug-flag settings (A/B of kernel variants on one box): FLAGS="0 2 256 ..."
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 SRF_STREAM_SPLIT=${SPLIT:-off}
for f in ${FLAGS:-0}; do
  timeout 200 python bench.py --workload ${W:-cfg2_improved_u16} --steps 20 --warmup 3 --no-cpu-baseline --debug-flags $f 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
ks = d.get('kernels', {})
print('flags %-10s ms/step %.3f  ' % ('$f', d['ms_per_step']) + '  '.join('%s=%.0f' % (k.replace('pw_conv_','').replace('pyramid_','py_'), v['avg_launch_us']) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]['ms_per_forward'])[:8]), 'check', d['self_check']['ok'])
"
done
