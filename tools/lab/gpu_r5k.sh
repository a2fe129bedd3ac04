#!/bin/bash
# (tools/gpu_r5k.sh: the debug switch it toggles -- paired-block GEMM with one tile per block, 1 << 20 -- was a one-line experiment, not kept)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for w in cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do for rep in 1 2; do for fl in 0 1048576; do
  timeout 300 python bench.py --workload $w --steps 12 --warmup 6 --no-cpu-baseline --no-kernel-profile --debug-flags $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w flags $fl -> %s  %.3f ms (median %.3f) ok=%s'%(d['config']['stream_split'], d['ms_per_step'], d['step_ms']['median'], d['self_check']['ok']))"
done; done; done
