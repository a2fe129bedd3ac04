#!/bin/bash
# Runs ON the GPU box: same-box A/B of bench.py over debug-flag values (include/sudormrf_hip.h, SRF_DIAGNOSTICS), alternating.
#   tools/gpu_ab.sh <out dir under gpurun_out> "<workloads>" "<flag values>" [reps] [extra bench args]
# e.g. the round-3 comparisons:  fused tail     tools/gpu_ab.sh ab "cfg2_improved_u16 cfg5_improved_u36_n4096" "0 32768" 2
#                                narrow tiles   tools/gpu_ab.sh ab "cfg1_improved_u8" "0 2048" 2 "--steps 200 --warmup 20"
#                                m-tile groups  tools/gpu_ab.sh ab "cfg5_improved_u36_n4096" "0 2" 2
# A previous build can be compared through SRF_LIB=<path to its .so> in the environment of a second call.
set -u
OUT=gpurun_out/${1:-ab}; W=${2:-cfg2_improved_u16}; F=${3:-0}; REPS=${4:-2}; EXTRA=${5:---steps 20 --warmup 4}
mkdir -p "$OUT"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for w in $W; do
  for rep in $(seq 1 $REPS); do
    for f in $F; do
      timeout 600 python bench.py --workload $w $EXTRA --no-cpu-baseline --debug-flags $f > "$OUT/${w}_f${f}_$rep.json" 2> "$OUT/${w}_f${f}_$rep.err"
      python - "$OUT/${w}_f${f}_$rep.json" "$w flags=$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks = d.get("kernels", {})
    top = sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_forward"])[:8]
    print(sys.argv[2], "ms", round(d["ms_per_step"], 4), "value", round(d["value"]), {k: (v["launches_per_forward"], round(v["avg_launch_us"], 1)) for k, v in top})
except Exception as e:
    print("bench failed", sys.argv[1], e)
PY
    done
  done
done
