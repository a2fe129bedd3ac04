#!/bin/bash
# Runs ON the GPU box (via gpurun): smoke -> GPU tests -> bench -> rocprofv3 kernel trace.
# Everything of interest is written under gpurun_out/ (merged back into the repo's gpurun_out/).
set -u
OUT=gpurun_out/${1:-r01}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
STEPS=${STEPS:-30}
{
  echo "== env"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4; nproc; 
  python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))"
} > "$OUT/env.log" 2>&1
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"
tail -5 "$OUT/smoke.log"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== pytest -m gpu"
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rA ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
  grep -E "passed|failed|error" "$OUT/pytest_gpu.log" | tail -3
  grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head -40
fi
echo "== bench"
timeout 600 python bench.py --steps $STEPS --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
for km in ${EXTRA_MODES:-2}; do
  timeout 600 python bench.py --steps 10 --warmup 3 --kernel-mode $km --no-cpu-baseline > "$OUT/bench_mode$km.json" 2> "$OUT/bench_mode$km.err"
  cat "$OUT/bench_mode$km.json"
done
for w in ${EXTRA_WORKLOADS:-}; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
  cat "$OUT/bench_$w.json"; tail -2 "$OUT/bench_$w.err"
done
for w in ${TRAIN_WORKLOADS:-}; do
  timeout 600 python tools/train_bench.py $w 32 5 > "$OUT/train_$w.json" 2> "$OUT/train_$w.err"
  python - "$OUT/train_$w.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: v for k, v in d.items() if k != "kernels_ms"})
PY
done
if [ "${SKIP_PROF:-0}" != "1" ]; then
  echo "== rocprofv3 kernel trace (single stream: the per-kernel durations bench.py's roofline block reports)"
  ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile ) > "$OUT/rocprof.log" 2>&1
  echo "rocprof rc=$?"
  f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160
  if [ "${SKIP_PROF2S:-0}" != "1" ]; then
  echo "== rocprofv3 kernel trace (default: auto-tuned two-stream split)"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof2s" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile ) > "$OUT/rocprof2s.log" 2>&1
  echo "rocprof(2 streams) rc=$?"
  fi
  # keep the merged-back payload small: the raw kernel traces can be large
  find "$OUT/prof" "$OUT/prof2s" -name "*kernel_trace.csv" -delete
fi
# PMC passes: counters in their own runs (kernel-trace only), one counter group per pass
i=0
for grp in ${PMC_GROUPS:-}; do
  i=$((i+1))
  echo "== rocprofv3 pmc pass $i: $grp"
  ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc ${grp//,/ } -d "$GRAFT_REPO_ROOT/$OUT/pmc$i" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile ) > "$OUT/pmc$i.log" 2>&1
  echo "pmc rc=$?"; ls "$OUT/pmc$i" | head -5
  find "$OUT/pmc$i" -name "*.csv" -size +30M -delete
done
echo "== done"
