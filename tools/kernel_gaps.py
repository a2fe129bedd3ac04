#!/usr/bin/env python3
"""How much of a step is the GPU idle BETWEEN kernels?  Reads a `rocprofv3 --kernel-trace` CSV of a single-stream run
(`bench.py --train ...`, or the forward with SRF_STREAM_SPLIT=off), takes the last `--steps` steps (a step starts at every
launch of `--first`, default the training step's first kernel), and reports per step: span, sum of kernel durations, launches,
idle time and the distribution of the gaps between consecutive kernels.

    python tools/kernel_gaps.py trace.csv [--first srf_x3w_pack_f16] [--steps 4]"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--first", default="srf_zero_kernel")
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if a.first in r[2]]
    # a step may launch `first` several times: keep the starts that are at least 5 ms apart
    marks = []
    for i in starts:
        if not marks or rows[i][0] - rows[marks[-1]][0] > 5_000_000:
            marks.append(i)
    if len(marks) < a.steps + 1:
        raise SystemExit("only %d steps found" % (len(marks) - 1))
    lo, hi = marks[-a.steps - 1], marks[-1]
    seg = rows[lo:hi]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
    n = a.steps
    print("%d steps: span %.3f ms per step, kernels %.3f ms per step (%d launches per step), idle between kernels %.3f ms per step = %.1f %%"
          % (n, span / n * 1e-6, busy / n * 1e-6, len(seg) // n, sum(gaps) / n * 1e-6, 100.0 * sum(gaps) / span))
    gs = sorted(gaps)
    print("gap between consecutive kernels: median %.1f us, p90 %.1f us, p99 %.1f us, max %.1f us" %
          (gs[len(gs) // 2] * 1e-3, gs[int(0.9 * len(gs))] * 1e-3, gs[int(0.99 * len(gs))] * 1e-3, gs[-1] * 1e-3))
    big = sorted(((g, seg[i][2], seg[i + 1][2]) for i, g in enumerate(gaps)), reverse=True)[:12]
    print("largest gaps (us, after kernel -> before kernel):")
    for g, k0, k1 in big:
        print("   %8.1f  %-50s -> %s" % (g * 1e-3, k0.split("(")[0][-50:], k1.split("(")[0][-50:]))
    edges = [(1, "< 1 us"), (3, "1-3 us"), (6, "3-6 us"), (12, "6-12 us"), (50, "12-50 us"), (1e9, "> 50 us")]
    prev = 0
    for e, label in edges:
        sel = [g for g in gaps if prev * 1e3 <= g < e * 1e3]
        print("   %-9s %6d gaps, %.3f ms per step" % (label, len(sel) // n, sum(sel) / n * 1e-6))
        prev = e


if __name__ == "__main__":
    main()
