#!/bin/bash
# Round 3, first GPU call: smoke, the GPU suite (new parity tests included), GEMM A/B, bench A/B, cfg 4/5, train, feeder.
set -u
OUT=gpurun_out/${1:-r03a}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4; nproc; free -g | head -2; } > "$OUT/env.log" 2>&1
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -3 "$OUT/smoke.log"
echo "== gemm A/B"; timeout 600 python tools/gemm_ab.py > "$OUT/gemm_ab.log" 2>&1; echo "rc=$?"; grep -v "^{" "$OUT/gemm_ab.log" | tail -24
echo "== pytest -m gpu"
timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rA -s > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" "$OUT/pytest_gpu.log" | tail -3; grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head -40
echo "== bench A/B (cfg 2)"
for f in 0 16384 0 16384; do
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --debug-flags $f > "$OUT/bench_f$f.json" 2> "$OUT/bench_f$f.err"
  python - "$OUT/bench_f$f.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"], 3), "value", round(d["value"]), {k: round(v["avg_launch_us"], 1) for k, v in d.get("kernels", {}).items() if "x3v" in k or "pyramid" in k})
PY
done
for w in cfg4_improved_u36_n2048 cfg5_improved_u36_n4096 cfg3_groupcomm_u8 cfg1_improved_u8; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
  python - "$OUT/bench_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], "ms", round(d["ms_per_step"], 3), "value", round(d["value"]), "frac", round(d["forward_roofline"]["frac"], 3))
PY
done
echo "== train"; timeout 600 python bench.py --train --steps 5 --warmup 2 > "$OUT/train_cfg2.json" 2> "$OUT/train_cfg2.err"; tail -c 600 "$OUT/train_cfg2.json"
echo "== feeder"; timeout 600 python bench.py --feeder > "$OUT/feeder.json" 2> "$OUT/feeder.err"; cat "$OUT/feeder.json"; tail -3 "$OUT/feeder.err"
echo "== done"
