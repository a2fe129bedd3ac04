#!/bin/bash
# round 5: forward A/B of the fused conv pairs (debug flag 1 = separate launches) + the model tests
set -u
OUT=gpurun_out/${1:-r05d}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for f in 0 1; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --debug-flags $f > "$OUT/bench_cfg2_flags${f}_$rep.json" 2> "$OUT/bench_cfg2_flags${f}_$rep.err"; echo "bench flags=$f rep=$rep rc=$?"
  python - "$OUT/bench_cfg2_flags${f}_$rep.json" $rep <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  ms/step", d.get("ms_per_step"), "value", d.get("value"), "split", d.get("config",{}).get("stream_split"), "kernel_set frac", d.get("forward_roofline",{}).get("kernel_set",{}).get("frac"))
    if sys.argv[2] == "1":
        for k, v in d.get("kernels", {}).items(): print("    %-22s %2d x %7.1f us = %6.3f ms" % (k, v["launches_per_forward"], v["avg_launch_us"], v["ms_per_forward"]))
except Exception as e: print("  parse failed", e)
PY
done; done
SRF_STREAM_SPLIT=off timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single stream: ms/step', d['ms_per_step'])"
SRF_STREAM_SPLIT=off timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile --debug-flags 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single stream, flag 1: ms/step', d['ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_model.py -q > "$OUT/pytest_model.log" 2>&1; echo "pytest model rc=$?"; tail -5 "$OUT/pytest_model.log"
