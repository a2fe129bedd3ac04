#!/bin/bash
# Runs ON the GPU box: the cfg-2 forward under explicit stream splits (SRF_STREAM_SPLIT), two repetitions each.
set -u
OUT=gpurun_out/${1:-r05split}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
W=${2:-cfg2_improved_u16}
for rep in 1; do
  for sp in auto off half 5:3 3:1 11:5 9:7 21:11 2:1:1 5:3:0 3:3:2; do
    [ "$sp" = "5:3:0" ] && continue
    SRF_STREAM_SPLIT=$sp timeout 200 python bench.py --workload $W --steps 12 --warmup 6 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W split %-6s -> %s  %.3f ms  (median %.3f)'%('$sp', d['config']['stream_split'], d['ms_per_step'], d['step_ms']['median']))" | tee -a "$OUT/sweep_$W.txt"
  done
done
