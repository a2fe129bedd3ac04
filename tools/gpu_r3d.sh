#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03d}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export GEMM_ROUNDS=5 GEMM_ITERS=10
timeout 900 python tools/gemm_ab.py dyn=0 static=32768 x3v=16384 > "$OUT/dyn_ab.log" 2>&1
grep -v "^{" "$OUT/dyn_ab.log" | tail -30
echo "== pytest (model + ops)"
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider --tb=short -x > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -5 "$OUT/pytest_gpu.log"
for f in 0 32768 16384; do
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --debug-flags $f > "$OUT/bench_f$f.json" 2> "$OUT/bench_f$f.err"
  python - "$OUT/bench_f$f.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"], 3), "value", round(d["value"]), {k: round(v["avg_launch_us"], 1) for k, v in d.get("kernels", {}).items() if "x3v" in k or "pyramid" in k})
PY
done
echo "== done"
