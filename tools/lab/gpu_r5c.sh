#!/bin/bash
# round 5: pair kernel iteration -- parity, isolated A/B, timeline, forward A/B
set -u
OUT=gpurun_out/${1:-r05e}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "pair" > "$OUT/pytest_pair.log" 2>&1; echo "pytest pair rc=$?"; tail -4 "$OUT/pytest_pair.log"
PAIR_SHAPES=res timeout 300 python tools/pair_ab.py 32 20 12 > "$OUT/pair_ab.log" 2>&1; echo "pair_ab rc=$?"; grep -v "^{" "$OUT/pair_ab.log" | grep -v amdgpu.ids | tail -40
timeout 200 python tools/pair_timeline.py 32 12 > "$OUT/pair_timeline.log" 2>&1; echo "timeline rc=$?"; grep -v amdgpu.ids "$OUT/pair_timeline.log" | tail -12
for f in 0 1; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --debug-flags $f > "$OUT/bench_cfg2_flags${f}.json" 2> "$OUT/bench_cfg2_flags${f}.err"; echo "bench flags=$f rc=$?"
  python - "$OUT/bench_cfg2_flags${f}.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  ms/step", d.get("ms_per_step"), "value", d.get("value"), "split", d.get("config",{}).get("stream_split"), "kernel_set frac", d.get("forward_roofline",{}).get("kernel_set",{}).get("frac"))
    for k, v in d.get("kernels", {}).items():
        if k.startswith("pw_"): print("    %-22s %2d x %7.1f us = %6.3f ms" % (k, v["launches_per_forward"], v["avg_launch_us"], v["ms_per_forward"]))
except Exception as e: print("  parse failed", e)
PY
done
timeout 900 python -m pytest tests/test_gpu_model.py -q -x > "$OUT/pytest_model.log" 2>&1; echo "pytest model rc=$?"; tail -3 "$OUT/pytest_model.log"
