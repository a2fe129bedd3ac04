#!/bin/bash
# Runs ON the GPU box: same-box A/B of the training step between builds of the library (SRF_LIB; e.g. the previous commit
# built in a worktree and copied to tools/ab/libsudormrf_hip_base.so).  usage: train_lib_ab.sh OUTDIR "workload ..." name=path ...
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-train_ab}; mkdir -p "$OUT"
WL=${2:-cfg2_improved_u16}
shift 2
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  for wl in $WL; do
    for v in "$@"; do
      name=${v%%=*}; lib=${v#*=}; flags=0
      case "$lib" in *:*) flags=${lib#*:}; lib=${lib%%:*};; esac
      [ "$lib" = "-" ] && lib=$GRAFT_REPO_ROOT/sudo_rm_rf_amd/libsudormrf_hip.so
      SRF_LIB=$lib timeout 400 python "$GRAFT_REPO_ROOT/bench.py" --train --workload $wl --steps 10 --warmup 3 --no-cpu-baseline \
          --debug-flags $flags > "$OUT/${wl}_${name}_$rep.json" 2> "$OUT/${wl}_${name}_$rep.err"
      python - "$OUT/${wl}_${name}_$rep.json" "$wl $name $rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "FAILED", e); sys.exit(0)
ks = d.get("kernels", {})
top = sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"])[:12]
print("%-40s %.2f ms/step | " % (sys.argv[2], d["ms_per_step"]) + "  ".join("%s %.2f(%.0fus)" % (k, v["ms_per_step"], v["avg_launch_us"]) for k, v in top))
PY
    done
  done
done
