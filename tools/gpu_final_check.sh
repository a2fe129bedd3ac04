#!/bin/bash
# Runs ON the GPU box: what the driver runs at round end -- GPU tests, smoke(), the default bench line.
set -u
OUT=gpurun_out/${1:-r05check}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
[ "${2:-all}" = bench ] || { timeout 900 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -3 "$OUT/smoke.log"
SECONDS=0; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
echo "bench wall ${SECONDS}s"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("%.3f ms %.0f sep-s/s | %s %.1f us frac %.3f traffic %s | fwd %.3f kernel_set %.3f | cpu %s | host issue %.2f ms"%(d["ms_per_step"],d["value"],r["kernel"],r["avg_launch_us"],r["frac"],r.get("traffic"),d["forward_roofline"]["frac"],d["forward_roofline"]["kernel_set"]["frac"],d["cpu_baseline"]["value"],d["step_ms"]["host_issue_ms_per_step"]))
PY
