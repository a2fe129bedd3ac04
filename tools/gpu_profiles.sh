#!/bin/bash
# Runs ON the GPU box: the judged evidence of a round from the SHIPPED build (sudo_rm_rf_amd/libsudormrf_hip.so) -- GPU test summary,
# bench lines of all five configurations (+ the exact-fp32 line), the training steps (roofline + cpu_baseline), rocprofv3 kernel
# stats and PMC passes (HBM traffic; SQ MFMA / VALU / LDS counters) of the forward (cfgs 2, 4, 5) AND of the training step
# (cfgs 2, 4), package power.  No gate: whatever box the pool hands out (round 3's set came from boxes <= 6.95 ms).
# usage: tools/gpu_profiles.sh <out dir under gpurun_out> [parts: t b r p x (c = completion pass, after a first collect), default tbrpx]; then tools/collect_profiles.py <dir> [tag]
# parts: t = pytest -m gpu, b = bench lines, r = rocprofv3 kernel stats, p = PMC passes, x = round-5 extras (fused-pair A/B, HIP-event
# timeline of the two-stream forward, zero-vs-random operand probe)
set -u
OUT=gpurun_out/${1:-r05p}
PARTS=${2:-tbrpx}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4; nproc; date -u; } > "$OUT/env.log" 2>&1
if [[ $PARTS == *t* ]]; then
  timeout 1500 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
fi
if [[ $PARTS == *b* ]]; then
  timeout 900 python bench.py --steps 50 --warmup 10 > "$OUT/bench_cfg2_improved_u16.json" 2> "$OUT/bench_cfg2.err"; echo "bench rc=$?"; tail -c 300 "$OUT/bench_cfg2_improved_u16.json"
  timeout 600 python bench.py --steps 20 --warmup 5 --kernel-mode 2 --no-cpu-baseline > "$OUT/bench_cfg2_exact_fp32.json" 2> "$OUT/bench_cfg2_exact.err"
  for w in cfg1_improved_u8 cfg3_groupcomm_u8 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do
    timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
  done
  timeout 900 python bench.py --train --steps 10 --warmup 3 > "$OUT/train_cfg2_improved_u16.json" 2> "$OUT/train_cfg2.err"; echo "train rc=$?"; tail -c 300 "$OUT/train_cfg2_improved_u16.json"
  # (cfg 3 with its cpu_baseline too -- VERDICT r5 next 8; cfg 4's CPU step does not fit the time budget)
  timeout 900 python bench.py --train --workload cfg3_groupcomm_u8 --steps 5 --warmup 2 > "$OUT/train_cfg3_groupcomm_u8.json" 2> "$OUT/train_cfg3_groupcomm_u8.err"
  timeout 600 python bench.py --train --workload cfg4_improved_u36_n2048 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/train_cfg4_improved_u36_n2048.json" 2> "$OUT/train_cfg4_improved_u36_n2048.err"
  for w in proj res_conv forward copy; do timeout 120 python tools/power_probe.py $w 3 2>/dev/null | tail -1 >> "$OUT/power.log"; done
fi
prof() {   # name, bench args...
  local name=$1; shift
  ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_$name" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile ) > "$OUT/rocprof_$name.log" 2>&1
  find "$OUT/prof_$name" -name "*kernel_trace.csv" -delete
}
pmc() {    # name, index, counters (comma separated), bench args...
  local name=$1 i=$2 grp=$3; shift 3
  ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc ${grp//,/ } -d "$GRAFT_REPO_ROOT/$OUT/pmc_${name}_$i" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile ) > "$OUT/pmc_${name}_$i.log" 2>&1
  find "$OUT/pmc_${name}_$i" -name "*kernel_trace.csv" -delete
  find "$OUT/pmc_${name}_$i" -name "*.csv" -size +30M -delete
}
if [[ $PARTS == *r* ]]; then
  for w in cfg2_improved_u16 cfg3_groupcomm_u8 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do prof $w --workload $w; done
  for w in cfg2_improved_u16 cfg4_improved_u36_n2048; do prof train_$w --train --workload $w; done
fi
if [[ $PARTS == *p* ]]; then
  for w in cfg2_improved_u16 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do
    i=0
    for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_WAIT_INST_LDS,SQ_BUSY_CYCLES,SQ_INSTS_MFMA"; do
      i=$((i+1)); pmc $w $i "$grp" --workload $w
    done
  done
  for w in cfg2_improved_u16 cfg4_improved_u36_n2048; do
    i=0
    for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_WAIT_INST_LDS,SQ_BUSY_CYCLES,SQ_INSTS_MFMA"; do
      i=$((i+1)); pmc train_$w $i "$grp" --train --workload $w
    done
  done
fi
if [[ $PARTS == *x* ]]; then
  timeout 300 python tools/pair_ab.py 32 20 12 > "$OUT/pair_ab.log" 2>&1
  timeout 300 python tools/two_stream_events.py --forwards 10 --json "$OUT/two_stream_events.json" > "$OUT/two_stream_events.txt" 2>&1
  timeout 200 python tools/pair_power_probe.py 32 > "$OUT/pair_power_probe.log" 2>&1
fi
if [[ $PARTS == *c* ]]; then   # completion pass: cfg-3 counters (forward + training step), cfg-4 training step WITH its CPU baseline, and the
  # cfg-2 lines again now that profiles/ holds this round's PMC traffic (bench.py reads it from there)
  for w in cfg3_groupcomm_u8; do
    prof train_$w --train --workload $w
    i=0
    for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_WAIT_INST_LDS,SQ_BUSY_CYCLES,SQ_INSTS_MFMA"; do
      i=$((i+1)); pmc $w $i "$grp" --workload $w; pmc train_$w $i "$grp" --train --workload $w
    done
  done
  timeout 900 python bench.py --train --workload cfg4_improved_u36_n2048 --steps 5 --warmup 2 > "$OUT/train_cfg4_improved_u36_n2048.json" 2> "$OUT/train_cfg4.err"
  timeout 900 python bench.py --steps 50 --warmup 10 > "$OUT/bench_cfg2_improved_u16.json" 2> "$OUT/bench_cfg2.err"
  timeout 900 python bench.py --train --steps 10 --warmup 3 > "$OUT/train_cfg2_improved_u16.json" 2> "$OUT/train_cfg2.err"
fi
if [[ $PARTS == *R* ]]; then   # the training step's kernel stats and counter passes again (after a change to its kernel set)
  for w in cfg2_improved_u16 cfg4_improved_u36_n2048; do
    prof train_$w --train --workload $w
    i=0
    for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_WAIT_INST_LDS,SQ_BUSY_CYCLES,SQ_INSTS_MFMA"; do
      i=$((i+1)); pmc train_$w $i "$grp" --train --workload $w
    done
  done
fi
if [[ $PARTS == *T* ]]; then   # the three training lines again (after a change to the training step)
  timeout 900 python bench.py --train --steps 10 --warmup 3 > "$OUT/train_cfg2_improved_u16.json" 2> "$OUT/train_cfg2.err"
  timeout 600 python bench.py --train --workload cfg3_groupcomm_u8 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/train_cfg3_groupcomm_u8.json" 2> "$OUT/train_cfg3.err"
  timeout 900 python bench.py --train --workload cfg4_improved_u36_n2048 --steps 5 --warmup 2 > "$OUT/train_cfg4_improved_u36_n2048.json" 2> "$OUT/train_cfg4.err"
fi
echo "== done"; du -sh "$OUT"
