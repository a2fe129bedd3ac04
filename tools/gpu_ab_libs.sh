#!/bin/bash
# Runs ON the GPU box: same-box A/B of library builds (gpurun_ab_<name>.so at the repo root, selected through SRF_LIB)
# with bench.py's own per-kernel HIP-event profile (no rocprof): VARIANTS="base x y", REPS=2, W=workload, SPLIT=off|auto
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 SRF_STREAM_SPLIT=${SPLIT:-off}
for rep in $(seq 1 ${REPS:-2}); do
  for v in ${VARIANTS:-base new}; do
    SRF_LIB=$GRAFT_REPO_ROOT/gpurun_ab_$v.so timeout 200 python bench.py --workload ${W:-cfg2_improved_u16} --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
ks = d.get('kernels', {})
print('%-10s rep $rep ms/step %.3f  ' % ('$v', d['ms_per_step']) + '  '.join('%s=%.1f' % (k.replace('pw_conv_','').replace('pyramid_','py_'), v['avg_launch_us']) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]['ms_per_forward'])[:9]), 'check', d['self_check']['ok'])
"
  done
done
