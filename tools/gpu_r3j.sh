#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03j}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/gemm_timeline.py > "$OUT/timeline.log" 2>&1; echo rc=$?
grep -v "^{" "$OUT/timeline.log" | tail -30
echo "== epilogue without stores (proj only)"
TL_EXTRA=$((1<<30)) timeout 600 python tools/gemm_timeline.py > "$OUT/timeline_nostores.log" 2>&1; echo rc=$?
grep -v "^{" "$OUT/timeline_nostores.log" | head -14
