#!/bin/bash
# Runs ON the GPU box: the judged evidence of one round -- bench lines of all five configurations, the training steps, rocprofv3
# kernel stats and the PMC passes (HBM traffic; SQ MFMA / VALU / LDS counters) for cfgs 2, 4 and 5, package power while the GEMMs run.
# usage: tools/gpu_profiles.sh <out dir under gpurun_out> ; then tools/collect_profiles3.sh <dir> <tag>
set -u
OUT=gpurun_out/${1:-r03p}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4; nproc; } > "$OUT/env.log" 2>&1
timeout 900 python bench.py --steps 50 --warmup 10 > "$OUT/bench_cfg2_improved_u16.json" 2> "$OUT/bench_cfg2.err"; echo "bench rc=$?"; tail -c 400 "$OUT/bench_cfg2_improved_u16.json"
for w in cfg1_improved_u8 cfg3_groupcomm_u8 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
for w in cfg2_improved_u16 cfg3_groupcomm_u8 cfg4_improved_u36_n2048; do
  timeout 600 python bench.py --train --workload $w --steps 5 --warmup 2 > "$OUT/train_$w.json" 2> "$OUT/train_$w.err"
done
for w in proj res_conv forward copy; do timeout 120 python tools/power_probe.py $w 3 2>/dev/null | tail -1 >> "$OUT/power.log"; done
for w in cfg2_improved_u16 cfg4_improved_u36_n2048 cfg5_improved_u36_n4096; do
  ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_$w" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile ) > "$OUT/rocprof_$w.log" 2>&1
  find "$OUT/prof_$w" -name "*kernel_trace.csv" -delete
  i=0
  for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_WAIT_INST_LDS,SQ_BUSY_CYCLES,SQ_INSTS_MFMA" "SQ_INSTS_VALU_MFMA_MOPS_BF16,SQ_WAVE_CYCLES,SQ_INSTS_LDS"; do
    i=$((i+1))
    ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc ${grp//,/ } -d "$GRAFT_REPO_ROOT/$OUT/pmc_${w}_$i" -o bench -- \
        python "$GRAFT_REPO_ROOT/bench.py" --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile ) > "$OUT/pmc_${w}_$i.log" 2>&1
    find "$OUT/pmc_${w}_$i" -name "*kernel_trace.csv" -delete
    find "$OUT/pmc_${w}_$i" -name "*.csv" -size +30M -delete
  done
done
echo "== done"; du -sh "$OUT"
