#!/bin/bash
set -u
OUT=gpurun_out/r02_diag2; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== tac conc (TW=2)"; R=12 timeout 300 python tools/diag_tac_conc.py > $OUT/tac_conc2.log 2>&1; echo rc=$?; grep -v "^    \|^       " $OUT/tac_conc2.log | cut -c1-200
