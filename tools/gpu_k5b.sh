#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03k5d}; mkdir -p "$OUT"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -p no:cacheprovider -x -k "decoder or overlap or fused_tail or separate or recipe" 2>&1 | tail -3
for w in cfg2_improved_u16 cfg5_improved_u36_n4096 cfg4_improved_u36_n2048; do
 for f in 0 32768; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline --debug-flags $f > "$OUT/bench_${w}_f$f.json" 2> "$OUT/bench_${w}_f$f.err"
  python - "$OUT/bench_${w}_f$f.json" $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["config"]["workload"][:24], "flags", sys.argv[2], "ms", round(d["ms_per_step"], 3), "value", round(d["value"]),
          {k: round(v["avg_launch_us"], 1) for k, v in d.get("kernels", {}).items() if "x3v<3>" in k or "mask" in k or "overlap" in k or "bf16x3" in k})
except Exception as e:
    print("bench failed", sys.argv[1], e)
PY
 done
done
