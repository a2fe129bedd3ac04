#!/bin/bash
# small-batch path: narrow-tile GEMM tests, cfg 1 (batch 1) and cfg 2 at batch 4 with / without it (flag 2048), eager / graph
set -u
OUT=gpurun_out/${1:-r03small}; mkdir -p "$OUT"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -x -k "narrow_tiles or encoder or persistent" 2>&1 | tail -30
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -p no:cacheprovider -x -k "golden or determinism" 2>&1 | tail -3
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
for f in 0 2048 0 2048; do
  timeout 300 python bench.py --workload cfg1_improved_u8 --steps 200 --warmup 20 --no-cpu-baseline --debug-flags $f > "$OUT/cfg1_f$f.json" 2> "$OUT/cfg1_f$f.err"; show "$OUT/cfg1_f$f.json" "cfg1 flags=$f"
done
SRF_GRAPH=always timeout 300 python bench.py --workload cfg1_improved_u8 --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/cfg1_graph.json" 2> "$OUT/cfg1_graph.err"; show "$OUT/cfg1_graph.json" "cfg1 graph"
for f in 0 2048 0 2048; do
  timeout 300 python bench.py --batch 4 --steps 100 --warmup 10 --no-cpu-baseline --debug-flags $f > "$OUT/cfg2b4_f$f.json" 2> "$OUT/cfg2b4_f$f.err"; show "$OUT/cfg2b4_f$f.json" "cfg2 batch4 flags=$f"
done
SRF_GRAPH=always timeout 300 python bench.py --batch 4 --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/cfg2b4_graph.json" 2> "$OUT/cfg2b4_graph.err"; show "$OUT/cfg2b4_graph.json" "cfg2 batch4 graph"
for b in 2 8; do
 for f in 0 2048; do
  timeout 300 python bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --debug-flags $f > "$OUT/cfg2b${b}_f$f.json" 2> "$OUT/cfg2b${b}_f$f.err"; show "$OUT/cfg2b${b}_f$f.json" "cfg2 batch$b flags=$f"
 done
done
