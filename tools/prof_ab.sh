#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel stats of bench.py for each debug-flag value given (A/B of kernel
# variants inside the full forward).  Usage: tools/prof_ab.sh <outdir> <flags...>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for f in "$@"; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$f" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" ${BENCH_ARGS:-} --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --debug-flags $f ) > "$OUT/rocprof_$f.log" 2>&1
  st=$(find "$OUT/prof_$f" -name "*kernel_stats.csv" | head -1)
  echo "== flags $f"; [ -n "$st" ] && cut -d, -f1-4 "$st" | head -14 | cut -c1-150
  find "$OUT/prof_$f" -name "*kernel_trace.csv" -delete
done
