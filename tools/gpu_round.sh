#!/bin/bash
# Runs ON the GPU box (via gpurun): ONE parametrised driver for a round's measurements (replaces the
# one-shot gpu_r5a.sh ... gpu_r5n.sh, now under tools/lab/).
#   tools/gpu_round.sh <out dir under gpurun_out> <part> [<part> ...]
# parts (each writes its own files under gpurun_out/<out dir>/):
#   smoke            __graft_entry__.smoke()
#   tests            pytest -m gpu (PYTEST_ARGS = extra arguments, e.g. "-k pair -x")
#   bench            the driver's bench line (cfg 2), STEPS / WARMUP from the environment (default 50 / 10)
#   benchall         bench lines of cfgs 1, 3, 4, 5 and the exact-fp32 line
#   train            bench.py --train for cfgs 2, 3, 4
#   prof             rocprofv3 --kernel-trace --stats of the single-stream cfg-2 forward (WORKLOADS overrides)
#   proftrain        the same for the training step (WORKLOADS, default cfg 2)
#   pmc / pmctrain   FETCH_SIZE, WRITE_SIZE and SQ counter passes (separate runs, kernel-trace only) for WORKLOADS
#   ab:<flags...>    same-box alternating A/B of bench.py over debug-flag values, e.g. ab:0,65536,131072 (WORKLOADS, REPS, AB_ARGS)
#   trainab:<flags>  the same for bench.py --train
#   pairab           tools/pair_ab.py (fused pair against the two launches; PAIR_FLAGS = extra variants)
#   l3               tools/l3_probe.py (Infinity-Cache probe), l3train: with the training step table
#   timeline         tools/two_stream_events.py (HIP-event timeline of the two-stream forward as timed)
#   power            tools/power_probe.py for the four standard loads
#   libab:pair|bench|train   same-box A/B of library BUILDS (gpurun_ab_<name>.so at the repo root; VARIANTS, default "r05 new")
#   py:<script>      python <script> (ARGS) with stdout / stderr kept
set -u
OUT=gpurun_out/${1:?out dir}; shift
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=${GRAFT_REPO_ROOT:-$PWD}
{ rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4; nproc; date -u; } > "$OUT/env.log" 2>&1
WL=${WORKLOADS:-cfg2_improved_u16}
PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_MFMA")

summ() {   # one-line summary of a bench JSON line
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks = d.get("kernels", {})
    key = "ms_per_forward" if ks and "ms_per_forward" in next(iter(ks.values())) else "ms_per_step"
    top = sorted(ks.items(), key=lambda kv: -kv[1][key])[:8]
    print(sys.argv[2], "ms", round(d["ms_per_step"], 4), "value", round(d["value"]),
          {k: (round(v.get("launches_per_forward", v.get("launches_per_step", 0))), round(v["avg_launch_us"], 1)) for k, v in top})
except Exception as e:
    print("bench failed", sys.argv[1], e)
PY
}
prof() {   # name, bench args...
  local name=$1; shift
  ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof_$name" -o bench -- \
      python "$ROOT/bench.py" "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile ) > "$OUT/rocprof_$name.log" 2>&1
  find "$OUT/prof_$name" -name "*kernel_trace.csv" -delete
  f=$(find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150
}
pmc() {    # name, bench args...
  local name=$1 i=0; shift
  for grp in "${PMC_GROUPS[@]}"; do
    i=$((i+1))
    ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$ROOT/$OUT/pmc_${name}_$i" -o bench -- \
        python "$ROOT/bench.py" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile ) > "$OUT/pmc_${name}_$i.log" 2>&1
    find "$OUT/pmc_${name}_$i" -name "*kernel_trace.csv" -delete
    find "$OUT/pmc_${name}_$i" -name "*.csv" -size +30M -delete
  done
}
ab() {     # "--train" or "", flags (comma separated)
  local mode=$1 flags=${2//,/ }
  for w in $WL; do
    for rep in $(seq 1 ${REPS:-2}); do
      for f in $flags; do
        timeout 600 python bench.py $mode --workload $w ${AB_ARGS:---steps 20 --warmup 5} --no-cpu-baseline --debug-flags $f \
          > "$OUT/ab${mode#--}_${w}_f${f}_$rep.json" 2> "$OUT/ab${mode#--}_${w}_f${f}_$rep.err"
        summ "$OUT/ab${mode#--}_${w}_f${f}_$rep.json" "$w ${mode#--} flags=$f rep=$rep"
      done
    done
  done
}

for part in "$@"; do
  echo "==== $part"
  case $part in
    smoke) timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -3 "$OUT/smoke.log" ;;
    tests) timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1
           echo "pytest rc=$?"; grep -E "passed|failed|error" "$OUT/pytest_gpu.log" | tail -3; grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head -30 ;;
    bench) timeout 900 python bench.py --steps ${STEPS:-50} --warmup ${WARMUP:-10} > "$OUT/bench_cfg2_improved_u16.json" 2> "$OUT/bench_cfg2.err"
           echo "bench rc=$?"; summ "$OUT/bench_cfg2_improved_u16.json" cfg2 ;;
    benchall)
      timeout 600 python bench.py --steps 20 --warmup 5 --kernel-mode 2 --no-cpu-baseline > "$OUT/bench_cfg2_exact_fp32.json" 2> "$OUT/bench_cfg2_exact.err"
      for w in cfg1_improved_u8 cfg3_groupcomm_u8 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do
        timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; summ "$OUT/bench_$w.json" $w
      done ;;
    train)
      timeout 900 python bench.py --train --steps 10 --warmup 3 > "$OUT/train_cfg2_improved_u16.json" 2> "$OUT/train_cfg2.err"; summ "$OUT/train_cfg2_improved_u16.json" "train cfg2"
      timeout 900 python bench.py --train --workload cfg3_groupcomm_u8 --steps 5 --warmup 2 > "$OUT/train_cfg3_groupcomm_u8.json" 2> "$OUT/train_cfg3.err"; summ "$OUT/train_cfg3_groupcomm_u8.json" "train cfg3"
      timeout 900 python bench.py --train --workload cfg4_improved_u36_n2048 --steps 5 --warmup 2 > "$OUT/train_cfg4_improved_u36_n2048.json" 2> "$OUT/train_cfg4.err"; summ "$OUT/train_cfg4_improved_u36_n2048.json" "train cfg4" ;;
    prof) for w in $WL; do prof $w --workload $w; done ;;
    proftrain) for w in $WL; do prof train_$w --train --workload $w; done ;;
    pmc) for w in $WL; do pmc $w --workload $w; done ;;
    pmctrain) for w in $WL; do pmc train_$w --train --workload $w; done ;;
    ab:*) ab "" "${part#ab:}" ;;
    trainab:*) ab --train "${part#trainab:}" ;;
    pairab) timeout 600 python tools/pair_ab.py ${PAIR_BATCHES:-32 20 12} > "$OUT/pair_ab.log" 2>&1; grep -v "^{" "$OUT/pair_ab.log" | tail -40 ;;
    l3) timeout 1500 python tools/l3_probe.py > "$OUT/l3_probe.txt" 2>&1; cat "$OUT/l3_probe.txt" ;;
    l3train) timeout 1500 python tools/l3_probe.py --train > "$OUT/l3_probe_train.txt" 2>&1; cat "$OUT/l3_probe_train.txt" ;;
    timeline) timeout 300 python tools/two_stream_events.py --forwards 10 --json "$OUT/two_stream_events.json" > "$OUT/two_stream_events.txt" 2>&1; tail -30 "$OUT/two_stream_events.txt" ;;
    power) for w in proj res_conv forward copy; do timeout 120 python tools/power_probe.py $w 3 2>/dev/null | tail -1 | tee -a "$OUT/power.log"; done ;;
    libab:*)   # same-box A/B of library builds: gpurun_ab_<name>.so at the repo root (VARIANTS, default "r05 new"; "new" = the shipped build)
      LIB=sudo_rm_rf_amd/libsudormrf_hip.so; [ -f gpurun_ab_new.so ] || cp $LIB gpurun_ab_new.so
      for rep in $(seq 1 ${REPS:-2}); do for v in ${VARIANTS:-r05 new}; do
        cp gpurun_ab_$v.so $LIB; echo "-- $v rep $rep"
        case ${part#libab:} in
          pair) PAIR_ROUNDS=3 timeout 300 python tools/pair_ab.py ${PAIR_BATCHES:-32 20} 2>&1 | grep -E " pair|two_x3p" | tee -a "$OUT/libab_pair_$v.log" ;;
          bench) timeout 600 python bench.py ${AB_ARGS:---steps 30 --warmup 6} --no-cpu-baseline > "$OUT/libab_bench_${v}_$rep.json" 2> "$OUT/libab_bench_${v}_$rep.err"; summ "$OUT/libab_bench_${v}_$rep.json" "$v" ;;
          train) for w in $WL; do timeout 600 python bench.py --train --workload $w ${AB_ARGS:---steps 8 --warmup 3} --no-cpu-baseline > "$OUT/libab_train_${w}_${v}_$rep.json" 2> "$OUT/libab_train_${w}_${v}_$rep.err"; summ "$OUT/libab_train_${w}_${v}_$rep.json" "$v $w"; done ;;
        esac
      done; done
      cp gpurun_ab_new.so $LIB ;;
    py:*) s=${part#py:}; n=$(basename "$s" .py); timeout ${PY_TIMEOUT:-900} python "$s" ${ARGS:-} > "$OUT/$n.log" 2> "$OUT/$n.err"; echo "rc=$?"; tail -${PY_TAIL:-40} "$OUT/$n.log"; tail -5 "$OUT/$n.err" ;;
    *) echo "unknown part $part" ;;
  esac
done
echo "== done"; du -sh "$OUT"
