#!/bin/bash
# Copy the judged artefacts of one gpu_round.sh run (gpurun_out/<dir>) into profiles/ (tracked).
# usage: tools/collect_profiles.sh <gpurun_out dir name> [round tag, default r01]
set -eu
R=gpurun_out/$1; T=${2:-r01}
cp $R/bench.json profiles/${T}_cfg2_bs32_bench.json
[ -s $R/bench_mode2.json ] && cp $R/bench_mode2.json profiles/${T}_cfg2_bs32_bench_exact_fp32_mfma.json
[ -s $R/bench_mode1.json ] && cp $R/bench_mode1.json profiles/${T}_cfg2_bs32_bench_generic_kernels.json
[ -s $R/bench_cfg1_improved_u8.json ] && cp $R/bench_cfg1_improved_u8.json profiles/${T}_cfg1_bs1_bench.json
[ -s $R/bench_cfg3_groupcomm_u8.json ] && cp $R/bench_cfg3_groupcomm_u8.json profiles/${T}_cfg3_groupcomm_bs32_bench.json
[ -s $R/bench_cfg4_improved_u36_n2048.json ] && cp $R/bench_cfg4_improved_u36_n2048.json profiles/${T}_cfg4_u36_n2048_bs32_bench.json
[ -s $R/bench_cfg5_improved_u36_n4096.json ] && cp $R/bench_cfg5_improved_u36_n4096.json profiles/${T}_cfg5_u36_n4096_8s16k_bs16_bench.json
for w in cfg2_improved_u16 cfg3_groupcomm_u8 cfg4_improved_u36_n2048; do
  [ -s $R/train_$w.json ] && cp $R/train_$w.json profiles/${T}_${w}_train_step_bs32.json
done
f=$(find $R/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::native" "$f" > profiles/${T}_cfg2_bs32_rocprofv3_kernel_stats.csv
f2=$(find $R/prof2s -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f2" ] && grep -v "at::native" "$f2" > profiles/${T}_cfg2_bs32_rocprofv3_kernel_stats_two_streams.csv
[ -d $R/pmc1 ] && python tools/pmc_summary.py $R profiles/${T}_cfg2_bs32_pmc_hbm_traffic.csv > /dev/null
grep -E "passed|failed" $R/pytest_gpu.log | tail -1 > profiles/${T}_pytest_gpu_summary.txt
cat $R/smoke.log | grep -E "smoke|build" >> profiles/${T}_pytest_gpu_summary.txt
ls -la profiles/
