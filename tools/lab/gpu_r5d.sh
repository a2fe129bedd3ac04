#!/bin/bash
# Runs ON the GPU box: full GPU test suite of the cleaned build + same-box A/B of the TAC forward (old library = SRF_LIB) + cfg-2 / cfg-3 bench.
set -u
OUT=gpurun_out/${1:-r05g}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
OLD=$GRAFT_REPO_ROOT/sudo_rm_rf_amd/libsudormrf_hip_old.so
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export SRF_LIB=$OLD; else unset SRF_LIB; fi
    timeout 300 python bench.py --workload cfg3_groupcomm_u8 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_cfg3_${lib}_$rep.json" 2> "$OUT/bench_cfg3_${lib}_$rep.err"
    python - "$OUT/bench_cfg3_${lib}_$rep.json" $lib <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get("kernels",{})
print(sys.argv[2], "cfg3 %.3f ms"%d["ms_per_step"], {n:round(v["avg_launch_us"],1) for n,v in k.items() if n.startswith(("tac","pw_conv_small"))})
PY
  done
done
unset SRF_LIB
timeout 400 python bench.py --steps 30 --warmup 5 > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"; tail -c 600 "$OUT/bench_cfg2.json" | head -c 300; echo
python - "$OUT/bench_cfg2.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg2 %.3f ms  %.0f sep-s/s  roofline %s %.3f  kernel_set %.3f  cpu %s"%(d["ms_per_step"],d["value"],d["roofline"]["kernel"],d["roofline"]["frac"],d["forward_roofline"]["kernel_set"]["frac"],d.get("cpu_baseline",{}).get("value")))
PY
