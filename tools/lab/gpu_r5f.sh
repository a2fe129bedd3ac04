#!/bin/bash
# Runs ON the GPU box: kernel-to-kernel idle time of the training step and of the single-stream forward (rocprofv3 kernel trace).
set -u
OUT=gpurun_out/${1:-r05i}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/tr_train" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --train --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-profile ) > "$OUT/tr_train.log" 2>&1
f=$(find "$OUT/tr_train" -name "*kernel_trace.csv" | head -1)
python tools/kernel_gaps.py "$f" --first "srf_x3w_pack_kernel<true>" --steps 4 | tee "$OUT/gaps_train.txt"
( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/tr_fwd" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile ) > "$OUT/tr_fwd.log" 2>&1
f=$(find "$OUT/tr_fwd" -name "*kernel_trace.csv" | head -1)
python tools/kernel_gaps.py "$f" --first srf_encoder_fast --steps 6 | tee "$OUT/gaps_fwd.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 fwd %.3f ms, host issue %.3f ms'%(d['ms_per_step'], d['step_ms']['host_issue_ms_per_step']))"
