#!/usr/bin/env python3
"""Summarise the SQ counter passes of tools/gpu_round.sh (pmc3.., one counter group per rocprofv3 pass) per kernel, with the
MFMA utilisation they imply.  usage: pmc_sq_summary.py <gpurun_out/rXX dir> [kernel_stats.csv]"""
import collections
import csv
import glob
import sys

R = sys.argv[1]
dur = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) / 1e3
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob(R + "/pmc*/bench_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "srf_" not in k or r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        k = k.split("(")[0].replace("void ", "")
        e = agg[k][r["Counter_Name"]]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
print("# rocprofv3 --kernel-trace --pmc <group> passes of `bench.py --steps 2 --warmup 1` (single stream), averages per launch.")
print("# SIMDs = 1024 (256 CUs x 4).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES is summed over")
print("# 32 shader engines (MI355X_MICROARCH.md): kernel cycles ~ SQ_BUSY_CYCLES / 32, MFMA utilisation = MFMA_BUSY / (1024 x that).")
for k, cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [1, 0])[1] * kv[1].get("SQ_BUSY_CYCLES", [1, 0])[0]):
    v = {c: x[1] / x[0] for c, x in cs.items()}
    n = max(x[0] for x in cs.values())
    line = "%-58s n=%-4d" % (k, n)
    if k in dur:
        line += " %7.1f us" % dur[k]
    print(line)
    print("    " + "  ".join("%s=%.4g" % (c, v[c]) for c in sorted(v)))
    if v.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        cyc = v["SQ_BUSY_CYCLES"] / 32
        print("    -> kernel ~%.0f k cycles; MFMA pipe busy %.1f %% of SIMD-cycles; VALU instr / SIMD = %.1f k; bf16 MFMA ops = %.3g" %
              (cyc / 1e3, 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), v.get("SQ_INSTS_VALU", 0) / 1024e3,
               v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0)))
