#!/bin/bash
# m-tile groups in the 256 x 128 GEMM's tile order: tests, then same-box A/B (groups | flag 2 = one group | previous build)
set -u
OUT=gpurun_out/${1:-r03grp}; mkdir -p "$OUT"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -p no:cacheprovider -x -k "persistent or x3v or fused_tail or bench_batch or mask" 2>&1 | tail -3
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks = d.get("kernels", {})
    print(sys.argv[2], "ms", round(d["ms_per_step"], 3), "value", round(d["value"]),
          {k: round(v["avg_launch_us"], 1) for k, v in ks.items() if "mask" in k or "x3v" in k})
except Exception as e:
    print("bench failed", sys.argv[1], e)
PY
}
for w in cfg5_improved_u36_n4096 cfg4_improved_u36_n2048 cfg2_improved_u16; do
 for rep in 1 2; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline > "$OUT/${w}_grp$rep.json" 2> "$OUT/${w}_grp$rep.err"; show "$OUT/${w}_grp$rep.json" "$w groups"
  timeout 600 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline --debug-flags 2 > "$OUT/${w}_one$rep.json" 2> "$OUT/${w}_one$rep.err"; show "$OUT/${w}_one$rep.json" "$w one-group"
  SRF_LIB=$PWD/sudo_rm_rf_amd/libsudormrf_prev.so timeout 600 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline > "$OUT/${w}_prev$rep.json" 2> "$OUT/${w}_prev$rep.err"; show "$OUT/${w}_prev$rep.json" "$w previous build"
 done
done
for f in 0 2; do
  ( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/$OUT/pmc_f$f" -o bench -- \
      python "$GRAFT_REPO_ROOT/bench.py" --workload cfg5_improved_u36_n4096 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --debug-flags $f ) > "$OUT/pmc_f$f.log" 2>&1
  find "$OUT/pmc_f$f" -name "*kernel_trace.csv" -delete
  python - "$OUT/pmc_f$f" $f <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "x3w_kernel<3" in r["Kernel_Name"] or "x3w_kernel<1" in r["Kernel_Name"]:
            e = agg[r["Kernel_Name"].split("(")[0]]; e[0] += 1; e[1] += float(r["Counter_Value"])
for k, (n, v) in agg.items(): print("flags", sys.argv[2], k, "launches", n, "FETCH_SIZE x2 MB/launch", round(2 * v / n / 1024, 1))
PY
  find "$OUT/pmc_f$f" -name "*.csv" -size +5M -delete
done
