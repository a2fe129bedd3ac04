#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03m}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -30 > "$OUT/smi_idle.txt"; cat "$OUT/smi_idle.txt" | head -30
for w in idle copy proj res_conv forward; do timeout 120 python tools/power_probe.py $w 3 2>&1 | tail -1 | tee -a "$OUT/power.log"; done
