#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -3
for w in cfg2_improved_u16 cfg3_groupcomm_u8 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do for rep in 1 2 3; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w auto -> %s  %.3f ms (median %.3f)'%(d['config']['stream_split'], d['ms_per_step'], d['step_ms']['median']))"
done; done
