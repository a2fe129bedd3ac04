# per-variant: cfg 4 forward, single stream, timing only, per-kernel profile
LIB=sudo_rm_rf_amd/libsudormrf_hip.so; cp $LIB gpurun_keep.so
for v in base mfma dma ld st bar frag dsw64 base; do
  cp gpurun_ab_$v.so $LIB
  SRF_BENCH_TIMING_ONLY=1 SRF_STREAM_SPLIT=off timeout 300 python bench.py --workload cfg4_improved_u36_n2048 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d['kernels']
print('$v', round(d['ms_per_step'],3), {k: round(v['avg_launch_us'],1) for k,v in ks.items() if 'x3p' in k or 'x3w' in k})"
done
cp gpurun_keep.so $LIB
