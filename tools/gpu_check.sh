#!/bin/bash
# quick GPU gate: GPU tests (+ optional -k filter in $K), bench cfg2 / cfg3
set -u
OUT=gpurun_out/${1:-check}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout ${TEST_TIMEOUT:-600} python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -x --timeout 120 ${K:+-k "$K"} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log | cut -c1-300
for w in ${WORKLOADS:-cfg2_improved_u16 cfg3_groupcomm_u8}; do
  timeout 240 python bench.py --workload $w --steps 30 --warmup 5 ${BENCH_ARGS:-} > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "bench $w rc=$?"
  python - $OUT/bench_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d[k] for k in ("value", "ms_per_step", "step_ms", "self_check") if k in d})
    print("  split", d["config"].get("stream_split"), "roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "avg_launch_us", "achieved")})
    ks = d.get("kernels", {})
    print("  " + "  ".join("%s=%.1fus x%d" % (k, v["avg_launch_us"], v["launches_per_forward"]) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_forward"])[:9]))
    if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cores"), "gpu/cpu", d.get("gpu_over_cpu"))
except Exception as e:
    print("bench parse failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
