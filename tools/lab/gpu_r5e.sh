#!/bin/bash
# Runs ON the GPU box: GPU tests of the current build + the training lines (host issue time; cfg-3 line with this round's PMC traffic).
set -u
OUT=gpurun_out/${1:-r05h}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
timeout 600 python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/train_cfg2_improved_u16.json" 2> "$OUT/train_cfg2.err"
timeout 600 python bench.py --train --workload cfg3_groupcomm_u8 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/train_cfg3_groupcomm_u8.json" 2> "$OUT/train_cfg3.err"
python - "$OUT" <<'PY'
import json,sys
for w in ("cfg2_improved_u16","cfg3_groupcomm_u8"):
    d=json.loads(open("%s/train_%s.json"%(sys.argv[1],w)).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(w, "%.2f ms"%d["ms_per_step"], "host issue %.2f ms"%d["host_issue_ms_per_step"], "dom", r["kernel"], "%.1f us"%r["avg_launch_us"], "frac %.3f"%r.get("frac",0), "traffic", r.get("traffic"),
          {k:round(v["avg_launch_us"]) for k,v in d["kernels"].items() if v.get("includes_host_gap")})
PY
