#!/bin/bash
set -u
OUT=gpurun_out/${1:-r04d}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp 2>/dev/null; cd - >/dev/null
rocprofv3 -L > "$OUT/counters.txt" 2>&1
grep -c . "$OUT/counters.txt"
grep -oE "^\s*(Name|Counter_Name)\s*:\s*\S+|\b(SQ|TA|TCP|TCC|TD|GRBM|SPI)_[A-Z0-9_a-z]+" "$OUT/counters.txt" | sed 's/.*://' | tr -d ' ' | sort -u > "$OUT/counter_names.txt"
wc -l "$OUT/counter_names.txt"
