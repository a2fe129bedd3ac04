#!/bin/bash
# fused tail (K5): the new tests, then bench A/B fused (0) vs materialised masked tensor (32768) on cfg 2 / 4 / 5
set -u
OUT=gpurun_out/${1:-r03k5}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -p no:cacheprovider --tb=short -rA -s -k "fused_tail or bench_batch or intermediates or determinism" > "$OUT/pytest_k5.log" 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" "$OUT/pytest_k5.log" | tail -3; grep -E "^(FAILED|ERROR)|max abs" "$OUT/pytest_k5.log" | head -40
for w in cfg2_improved_u16 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do
 for f in 0 32768 0 32768; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline --debug-flags $f > "$OUT/bench_${w}_f$f.json" 2> "$OUT/bench_${w}_f$f.err"
  python - "$OUT/bench_${w}_f$f.json" $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["config"]["workload"], "flags", sys.argv[2], "ms", round(d["ms_per_step"], 3), "value", round(d["value"]),
          {k: round(v["avg_launch_us"], 1) for k, v in d.get("kernels", {}).items() if "x3v<3>" in k or "mask" in k or "overlap" in k or "bf16x3" in k or "mfma" in k or "pack" in k or "transpose" in k})
except Exception as e:
    print("bench failed", sys.argv[1], e)
PY
 done
done
echo "== done"
