#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03graph}; mkdir -p "$OUT"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for g in off always off always; do
  SRF_GRAPH=$g timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile > "$OUT/cfg2_$g.json" 2> "$OUT/cfg2_$g.err"
  python - "$OUT/cfg2_$g.json" $g <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("cfg2 graph", sys.argv[2], round(d["ms_per_step"], 4), d["config"].get("stream_split"), d["config"].get("hip_graph_replay"))
except Exception as e: print("failed", e)
PY
done
for g in off always; do
  SRF_STREAM_SPLIT=off SRF_GRAPH=$g timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile > "$OUT/cfg2_nosplit_$g.json" 2> "$OUT/cfg2_nosplit_$g.err"
  python - "$OUT/cfg2_nosplit_$g.json" $g <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("cfg2 nosplit graph", sys.argv[2], round(d["ms_per_step"], 4), d["config"].get("stream_split"), d["config"].get("hip_graph_replay"))
except Exception as e: print("failed", e)
PY
done
