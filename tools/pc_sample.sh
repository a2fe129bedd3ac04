#!/bin/bash
# Runs ON the GPU box: stochastic PC sampling of the GEMM micro-benchmark (one shape), to see which
# instructions the wavefronts of the split-bf16 kernel sit on.  Output: gpurun_out/pcs/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pcs
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
export GEMM_SHAPES=${GEMM_SHAPES:-proj_1x1} GEMM_ITERS=${GEMM_ITERS:-200}
cd /tmp
timeout 240 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-method ${PCS_METHOD:-stochastic} \
  --pc-sampling-unit ${PCS_UNIT:-cycles} --pc-sampling-interval ${PCS_INTERVAL:-1048576} \
  --kernel-trace --output-format csv -d "$OUT" -o pcs -- python "$GRAFT_REPO_ROOT/tools/gemm_bench.py" 0u \
  > "$OUT/run.log" 2>&1
echo "rc=$?" >> "$OUT/run.log"
tail -5 "$OUT/run.log"
ls -la "$OUT" | head
# keep the payload small: aggregate samples per (instruction, stall reason) on the box
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, os
out = sys.argv[1]
for f in glob.glob(out + "/**/*pc_sampling*.csv", recursive=True):
    rows = csv.DictReader(open(f))
    agg = collections.Counter()
    n = 0
    cols = None
    for r in rows:
        cols = cols or list(r.keys())
        n += 1
        agg[tuple(r.get(k, "") for k in ("Instruction", "Instruction_Comment", "Wave_Issued_Instruction",
                                        "Instruction_Type", "Stall_Reason"))] += 1
    with open(f.replace(".csv", "_agg.txt"), "w") as g:
        g.write("columns: %s\nsamples: %d\n" % (cols, n))
        for k, v in agg.most_common(400):
            g.write("%7d  %s\n" % (v, " | ".join(k)))
    if os.path.getsize(f) > 8 << 20:
        os.remove(f)
PY
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
ls -la "$OUT"/* | head -20
