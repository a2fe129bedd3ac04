#!/usr/bin/env python3
"""Timeline of the forward AS IT IS TIMED: the engine runs a batch as two sub-batches on two HIP streams
(DESIGN.md "Two streams per forward"); every per-kernel number under profiles/ used to come from a single-stream pass.  This reads
a `rocprofv3 --kernel-trace` CSV of `bench.py` (no PMC) and reports, over the steady-state forwards at the end of the trace:

  * per stream: kernels, busy time (union of its kernel intervals), share of the span;
  * both streams: time with 2 / 1 / 0 kernels running (overlap, single, idle), the span per forward;
  * per kernel family: launches per forward, average duration UNDER CO-RESIDENCY and, with a second (single-stream) trace, the
    same kernel's duration alone -- the stretch factor is what sharing the chip costs a kernel, the overlap what it buys.

    python tools/two_stream_timeline.py two_stream_kernel_trace.csv [single_stream_kernel_trace.csv] [--forwards 10]"""
import argparse
import csv
import re
import sys
from collections import defaultdict

FAMILY = [
    (r"srf_pw_x3f_kernel<1", "pw_pair_x3f<1> (bottleneck + proj_1x1)"), (r"srf_pw_x3f_kernel<2", "pw_pair_x3f<2> (res_conv + proj_1x1)"),
    (r"srf_pw_x3p_kernel<0", "pw_conv_x3p<0> (proj_1x1)"), (r"srf_pw_x3p_kernel<1", "pw_conv_x3p<1> (bottleneck)"),
    (r"srf_pw_x3p_kernel<2", "pw_conv_x3p<2> (res_conv)"), (r"srf_pw_x3w_kernel<3, 4", "pw_mask_decode (mask GEMM + decoder)"),
    (r"srf_pw_x3w_kernel", "pw_conv_x3w"), (r"srf_pyramid_reg_kernel<true", "pyramid_moments"),
    (r"srf_pyramid_reg_kernel<false", "pyramid_merge"), (r"srf_pyramid_finalize", "pyramid_finalize"),
    (r"srf_encoder", "encoder"), (r"srf_overlap_add", "overlap_add"), (r"srf_x3w_pack_kernel", "pack_pw_weights"),
    (r"srf_x3w_pack_dec", "pack_decoder"), (r"srf_zero_kernel", "zero_fill"), (r"srf_tac", "tac"), (r"srf_pw_small", "pw_conv_small"),
]


def family(name):
    for pat, fam in FAMILY:
        if re.search(pat, name):
            return fam
    m = re.search(r"(srf_\w+)", name)
    return m.group(1) if m else None


def load(path):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            fam = family(r["Kernel_Name"])
            if fam is None:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Queue_Id"), r.get("Stream_Id")), fam))
    rows.sort()
    return rows


def pick_forwards(rows, n, want_streams):
    """The last n forwards of the timed loop.  A forward starts at an encoder launch; encoder launches within 1 ms of each other
    belong to one forward (its sub-batches, one per stream).  The timed loop = the longest run of consecutive forwards that use
    `want_streams` streams (2: >= 2) and follow each other at a steady pace (period within 25 % of the run's median) -- the
    auto-tuner's trial forwards, bench.py's self-check forwards and host-side pauses fall outside it."""
    enc = [(r[0], r[2]) for r in rows if r[3] == "encoder"]
    if len(enc) < 2:
        return rows, 1
    fw = []                                              # [start, set of streams]
    for t, q in enc:
        if fw and t - fw[-1][0] < 1_000_000:
            fw[-1][1].add(q)
        else:
            fw.append([t, {q}])
    ok = [(len(f[1]) >= 2) if want_streams >= 2 else (len(f[1]) == 1) for f in fw]
    periods = sorted(fw[i + 1][0] - fw[i][0] for i in range(len(fw) - 1) if ok[i] and ok[i + 1])
    if not periods:
        return rows, 1
    med = periods[len(periods) // 2]
    best, cur = (0, 0), None
    for i in range(len(fw) - 1):
        steady = ok[i] and ok[i + 1] and abs((fw[i + 1][0] - fw[i][0]) - med) <= 0.25 * med
        if steady:
            cur = (cur[0], i + 1) if cur else (i, i + 1)
            if cur[1] - cur[0] > best[1] - best[0]:
                best = cur
        else:
            cur = None
    first, last = max(best[0], best[1] - n), best[1]     # forwards first .. last - 1, ended by the start of forward `last`
    lo, hi = fw[first][0], fw[last][0]
    return [r for r in rows if lo <= r[0] < hi], last - first


def union(intervals):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(intervals):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def depth_profile(rows):
    """ns with k kernels in flight, k = 0, 1, 2+ over the span of rows."""
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, out = 0, ev[0][0], defaultdict(int)
    for t, d in ev:
        out[min(depth, 2)] += t - last
        last = t
        depth += d
    return out


def summarise(rows, nfw, label, ref=None):
    span = max(r[1] for r in rows) - min(r[0] for r in rows)
    print("== %s: %d forward(s), %.3f ms per forward (first launch of the window to the last completion)" % (label, nfw, span / nfw * 1e-6))
    streams = sorted({r[2] for r in rows})
    for st in streams:
        mine = [(s, e) for s, e, q, _ in rows if q == st]
        print("   stream (queue %s, stream %s): %5d kernels, busy %.3f ms per forward = %4.1f %% of the span" %
              (st[0], st[1], len(mine), union(mine) / nfw * 1e-6, 100.0 * union(mine) / span))
    dp = depth_profile(rows)
    print("   kernels in flight: two or more %.1f %%, one %.1f %%, none (gaps) %.1f %% of the span" %
          (100.0 * dp[2] / span, 100.0 * dp[1] / span, 100.0 * dp[0] / span))
    fam = defaultdict(list)
    for s, e, _, f in rows:
        fam[f].append(e - s)
    tot = sum(sum(v) for v in fam.values())
    print("   %-42s %9s %12s %12s %9s" % ("kernel family", "per fwd", "avg us", "ms per fwd", "alone us" if ref else ""))
    for f, v in sorted(fam.items(), key=lambda kv: -sum(kv[1])):
        alone = ""
        if ref and f in ref:
            alone = "%9.1f (x %.2f)" % (ref[f], (sum(v) / len(v) * 1e-3) / ref[f])
        print("   %-42s %9.1f %12.1f %12.3f %s" % (f, len(v) / nfw, sum(v) / len(v) * 1e-3, sum(v) / nfw * 1e-6, alone))
    print("   sum of kernel durations %.3f ms per forward = %.2f x the span (> 1: the streams overlap)" % (tot / nfw * 1e-6, tot / span))
    return {f: sum(v) / len(v) * 1e-3 for f, v in fam.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("two_stream")
    ap.add_argument("single_stream", nargs="?")
    ap.add_argument("--forwards", type=int, default=10)
    a = ap.parse_args()
    ref = None
    if a.single_stream:
        rows, nfw = pick_forwards(load(a.single_stream), a.forwards, 1)
        ref = summarise(rows, nfw, "single stream (SRF_STREAM_SPLIT=off)")
    rows, nfw = pick_forwards(load(a.two_stream), a.forwards, 2)
    summarise(rows, nfw, "two streams (the timed configuration)", ref)


if __name__ == "__main__":
    sys.exit(main())
