#!/usr/bin/env python3
"""Per kernel: averages per launch of every counter collected by tools/gpu_gemm_pmc.sh, plus durations from the kernel traces.
usage: gemm_pmc_summary.py <gpurun_out dir> [name filter]"""
import collections
import csv
import glob
import sys

R = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "srf_pw_x3"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(R + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if flt not in k:
            continue
        e = agg[k][r["Counter_Name"]]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
for f in sorted(glob.glob(R + "/pmc_*/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if flt in k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    d = sorted(dur.get(k, [0]))
    print("%s  n=%d  median %.1f us" % (k, len(d), d[len(d) // 2]))
    for c in sorted(agg[k]):
        n, s = agg[k][c]
        print("    %-34s %.5g" % (c, s / n))
