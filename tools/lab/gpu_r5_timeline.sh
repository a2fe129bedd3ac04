#!/bin/bash
# round 5: rocprofv3 kernel trace (no PMC) of the cfg-2 forward AS TIMED (two streams) and single-stream -> tools/two_stream_timeline.py
set -u
OUT=gpurun_out/${1:-r05tl}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in two single; do
  if [ $mode = single ]; then export SRF_STREAM_SPLIT=off; else unset SRF_STREAM_SPLIT; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/trace_$mode" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --no-kernel-profile ) > "$OUT/rocprof_$mode.log" 2>&1
  echo "$mode rc=$?"; tail -c 400 "$OUT/rocprof_$mode.log" | grep -o '"ms_per_step": [0-9.]*'
done
unset SRF_STREAM_SPLIT
T2=$(find "$OUT/trace_two" -name "*kernel_trace.csv" | head -1); T1=$(find "$OUT/trace_single" -name "*kernel_trace.csv" | head -1)
python tools/two_stream_timeline.py "$T2" "$T1" --forwards 10 > "$OUT/two_stream_timeline.txt" 2>&1; cat "$OUT/two_stream_timeline.txt"
# keep the traces small: only our kernels' rows of the analysed window are needed for the summary
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
