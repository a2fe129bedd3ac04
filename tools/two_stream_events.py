#!/usr/bin/env python3
"""Timeline of the forward AS IT IS TIMED -- two sub-batches on two HIP streams (DESIGN.md "Two streams per forward") -- from HIP
events recorded by the in-library profiler on whichever stream a kernel was launched on (srf_profile_timeline).  rocprofv3's
kernel trace cannot show this: under the profiler the two queues take turns (profiles/r05_cfg2_rocprofv3_serialises_streams.txt),
so the engine's auto-tuner even picks the un-split forward there.  An event after every launch costs a little (the un-instrumented
forward of the same process is printed beside it), but it does not order the streams against each other.

Per forward, over the steady state: span, per-stream busy time, time with two / one / no kernel in flight, and per kernel family
the average duration UNDER CO-RESIDENCY beside the same kernel's duration in a single-stream pass of the same process.

    [SRF_STREAM_SPLIT=half|5:3] python tools/two_stream_events.py [--workload cfg2_improved_u16] [--forwards 10] [--json out.json]"""
import argparse
import ctypes as C
import json
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sudo_rm_rf_amd import _lib  # noqa: E402


def collect(lib, model, wav, nfw, dev):
    """[(name, t_end_ms, stream)] of nfw forwards, and the instrumented time per forward."""
    stream = _lib.current_stream(dev)
    torch.cuda.synchronize(dev)
    lib.srf_profile_begin(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(nfw):
            model(wav)
    e1.record()
    torch.cuda.synchronize(dev)
    cnt = C.c_int(0)
    _lib.check(lib.srf_profile_end(stream, C.byref(cnt)), "srf_profile_end")
    name, t, st = C.c_char_p(), C.c_float(), C.c_int()
    marks = []
    for i in range(cnt.value):
        _lib.check(lib.srf_profile_timeline(i, C.byref(name), C.byref(t), C.byref(st)), "srf_profile_timeline")
        marks.append((name.value.decode(), t.value, st.value))
    return marks, e0.elapsed_time(e1) / nfw


def intervals(marks):
    """Per stream, a kernel's interval = [previous mark on that stream (or the forward's first mark), its own mark]."""
    last = {}
    out = []
    t_begin = 0.0
    for name, t, st in marks:
        if name.startswith("("):            # "(gap)": start of a forward on this stream
            last[st] = t
            continue
        s = last.get(st, t_begin)
        out.append((s, t, st, name))
        last[st] = t
    return out


def union(iv):
    tot, cs, ce = 0.0, None, None
    for s, e in sorted(iv):
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ((ce - cs) if ce is not None else 0.0)


def depth(iv):
    ev = sorted([(s, 1) for s, _, _, _ in iv] + [(e, -1) for _, e, _, _ in iv])
    d, last, out = 0, ev[0][0], defaultdict(float)
    for t, k in ev:
        out[min(d, 2)] += t - last
        last, d = t, d + k
    return out


def report(label, iv, nfw, ms_fw, ref=None, parts=None):
    span = max(e for _, e, _, _ in iv) - min(s for s, _, _, _ in iv)
    res = {"forwards": nfw, "ms_per_forward_instrumented": ms_fw, "span_ms_per_forward": span / nfw, "streams": {}, "families": {}}
    print("== %s: %d forwards, %.3f ms per forward with the events in place" % (label, nfw, ms_fw))
    for st in sorted({x[2] for x in iv}):
        mine = [(s, e) for s, e, q, _ in iv if q == st]
        busy = union(mine)
        res["streams"][st] = {"kernels_per_forward": len(mine) / nfw, "busy_ms_per_forward": busy / nfw, "share_of_span": busy / span}
        print("   stream %d: %6.1f kernels per forward, busy %.3f ms per forward = %4.1f %% of the span" %
              (st, len(mine) / nfw, busy / nfw, 100 * busy / span))
    dp = depth(iv)
    res["in_flight"] = {"two": dp[2] / span, "one": dp[1] / span, "none": dp[0] / span}
    print("   kernels in flight: two %.1f %%, one %.1f %%, none %.1f %% of the span" % (100 * dp[2] / span, 100 * dp[1] / span, 100 * dp[0] / span))
    fam = defaultdict(list)
    for s, e, q, n in iv:
        fam[(n, q) if parts else (n, 0)].append(e - s)
    tot = sum(sum(v) for v in fam.values())
    # A sub-batch of b of the B examples does b / B of a kernel's work: "alone" = the single-stream duration of the whole batch
    # scaled by b / B (every kernel of the forward is linear in the batch at these sizes); stretch = co-resident / alone.
    print("   %-22s %6s %8s %10s %12s %s" % ("kernel family", "stream", "per fwd", "avg us", "ms per fwd",
                                              "alone us (whole batch x share) -> stretch" if ref else ""))
    for (n, q), v in sorted(fam.items(), key=lambda kv: -sum(kv[1])):
        avg = 1e3 * sum(v) / len(v)
        key = "%s@%d" % (n, q) if parts else n
        res["families"][key] = {"launches_per_forward": len(v) / nfw, "avg_us": avg, "ms_per_forward": sum(v) / nfw}
        extra = ""
        if ref and n in ref["families"]:
            share = parts[q] / float(sum(parts)) if parts and q < len(parts) else 1.0
            alone = ref["families"][n]["avg_us"] * share
            res["families"][key].update(alone_us=alone, stretch=avg / alone)
            extra = "%8.1f -> x %.2f" % (alone, avg / alone)
        print("   %-22s %6s %8.1f %10.1f %12.3f %s" % (n, q if parts else "-", len(v) / nfw, avg, sum(v) / nfw, extra))
    res["sum_of_kernel_ms_per_forward"] = tot / nfw
    print("   sum of kernel durations %.3f ms per forward = %.2f x the span" % (tot / nfw, tot / span))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2_improved_u16")
    ap.add_argument("--forwards", type=int, default=10)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as gc
    import sudo_rm_rf.dnn.models.improved_sudormrf as imp
    variant, kw, T, fs, batch = bench.WORKLOADS[a.workload]
    torch.manual_seed(0)
    model = (imp.SuDORMRF if variant == "improved" else gc.GroupCommSudoRmRf)(**kw).to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(1000)
    wav = torch.randn(batch, 1, T, generator=g)
    wav = ((wav - wav.mean(-1, keepdim=True)) / (wav.std(-1, keepdim=True) + 1e-9)).to(dev)
    eng = model._engine()
    with torch.no_grad():
        for _ in range(12):                      # warm-up + the auto-tuner's trials
            model(wav)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(a.forwards):
            model(wav)
    e1.record()
    torch.cuda.synchronize(dev)
    plain = e0.elapsed_time(e1) / a.forwards
    choice = eng._split_choice.get((dev.index, batch, T))
    print("workload %s batch %d: %.3f ms per forward un-instrumented, split %s" % (a.workload, batch, plain, choice))
    # single-stream reference pass of the same process
    was = eng.multi_stream
    eng.multi_stream = False
    with torch.no_grad():
        model(wav)
    m1, ms1 = collect(lib, model, wav, a.forwards, dev)
    eng.multi_stream = was
    ref = report("single stream", intervals(m1), a.forwards, ms1)
    with torch.no_grad():
        model(wav)
    m2, ms2 = collect(lib, model, wav, a.forwards, dev)
    two = report("two streams (the timed configuration, split %s)" % (choice,), intervals(m2), a.forwards, ms2, ref,
                 parts=list(choice) if choice and len(choice) > 1 else None)
    out = {"workload": a.workload, "batch": batch, "ms_per_forward_uninstrumented": plain, "split": choice, "single_stream": ref,
           "two_streams": two}
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
