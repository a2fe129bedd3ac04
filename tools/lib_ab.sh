#!/bin/bash
# Runs ON the GPU box: same-box A/B of several builds of the library (gpurun_ab_<name>.so at the repo root, e.g. the
# previous commit built in a worktree; names in $VARIANTS, default "old new"): alternating single-stream rocprof
# kernel stats + bench lines.  The last variant is left installed.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-lib_ab}; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=$GRAFT_REPO_ROOT/sudo_rm_rf_amd/libsudormrf_hip.so
for rep in 1 2; do
  for v in ${VARIANTS:-old new}; do
    cp $GRAFT_REPO_ROOT/gpurun_ab_$v.so $LIB
    ( cd /tmp && SRF_STREAM_SPLIT=off timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${v}_$rep" -o bench -- \
        python "$GRAFT_REPO_ROOT/bench.py" ${BENCH_ARGS:-} --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile ) > "$OUT/rocprof_${v}_$rep.log" 2>&1
    find "$OUT/prof_${v}_$rep" -name "*kernel_trace.csv" -delete
    echo "== $v $rep"
    python - "$OUT/prof_${v}_$rep/bench_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("  ".join("%s=%.1f" % (r["Name"].split("(")[0][-28:], float(r["AverageNs"]) / 1e3) for r in rows[:6]))
PY
    python "$GRAFT_REPO_ROOT/bench.py" ${BENCH_ARGS:-} --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-profile | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   two-stream ms/step', round(d['ms_per_step'],3))"
  done
done
last=$(echo ${VARIANTS:-old new} | awk '{print $NF}'); cp $GRAFT_REPO_ROOT/gpurun_ab_$last.so $LIB
