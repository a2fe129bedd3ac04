#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03small3}; mkdir -p "$OUT"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -x -k "narrow_tiles or encoder or persistent" 2>&1 | tail -4
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks = d.get("kernels", {})
    print(sys.argv[2], "ms", round(d["ms_per_step"], 4), "value", round(d["value"]),
          {k: (v["launches_per_forward"], round(v["avg_launch_us"], 1)) for k, v in ks.items() if v["ms_per_forward"] > 0.02})
except Exception as e:
    print("bench failed", sys.argv[1], e)
PY
}
for b in 1 2 4; do
 for f in 0 2048 0 2048; do
  timeout 300 python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --debug-flags $f > "$OUT/cfg2b${b}_f$f.json" 2> "$OUT/cfg2b${b}_f$f.err"; show "$OUT/cfg2b${b}_f$f.json" "cfg2 batch$b flags=$f"
 done
done
for f in 0 2048; do
  timeout 300 python bench.py --workload cfg1_improved_u8 --steps 200 --warmup 20 --no-cpu-baseline --debug-flags $f > "$OUT/cfg1_f$f.json" 2> "$OUT/cfg1_f$f.err"; show "$OUT/cfg1_f$f.json" "cfg1 flags=$f"
  timeout 300 python bench.py --workload cfg3_groupcomm_u8 --batch 1 --steps 100 --warmup 20 --no-cpu-baseline --debug-flags $f > "$OUT/cfg3b1_f$f.json" 2> "$OUT/cfg3b1_f$f.err"; show "$OUT/cfg3b1_f$f.json" "cfg3 batch1 flags=$f"
done
