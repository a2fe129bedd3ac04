#!/usr/bin/env python3
"""Co-residency probe (round 6): the fused conv pair on one stream NEXT TO the fused pyramid on another stream, both launched together --
how much does each stretch?  And the block period of both streams running pair, pyramid, pair, ... free or phase-locked with events
(each pair launch waits for the other stream's latest pair launch).  Decided against an anti-phase schedule of the two-stream forward
(profiles/r06_two_stream_phase_lock_ab.txt; the withdrawn library side is tools/lab/phase_lock_gate.patch).
    python tools/corun_probe.py [Bt per stream, default 16]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"


def main():
    Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    L, K1, Cmid, C2, D = 3200, 512, 256, 512, 5
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(Bt, K1, L, generator=g, device=DEV) * 1.3 + 0.2
    w1 = torch.randn(Cmid, K1, 1, generator=g, device=DEV) * K1 ** -0.5
    b1 = torch.randn(Cmid, generator=g, device=DEV)
    w2 = torch.randn(C2, Cmid, 1, generator=g, device=DEV) * Cmid ** -0.5
    b2 = torch.randn(C2, generator=g, device=DEV)
    res = torch.randn(Bt, Cmid, L, generator=g, device=DEV)
    slope = torch.tensor([0.17], device=DEV)
    gamma, beta = torch.rand(K1, generator=g, device=DEV) + 0.5, torch.randn(K1, generator=g, device=DEV) * 0.3
    sums = ops.gln_stats(x, Bt)
    p1, p2 = ops.pack_pw_weight(w1), ops.pack_pw_weight(w2)
    y1 = torch.randn(Bt, C2, L, generator=g, device=DEV)
    ysums = ops.gln_stats(y1, Bt)
    ws = [torch.randn(C2, 1, 5, generator=g, device=DEV) * 0.4 for _ in range(D)]
    bs = [torch.randn(C2, generator=g, device=DEV) * 0.1 for _ in range(D)]
    gs = [torch.rand(C2, generator=g, device=DEV) + 0.5 for _ in range(D)]
    be = [torch.randn(C2, generator=g, device=DEV) * 0.3 for _ in range(D)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def pair():
        return ops.pw_conv_pair(x, p1, b1, sums, gamma, beta, slope, res, p2, b2, Cmid, C2)

    def pyr():
        return ops.pyramid(y1, ysums, gamma, beta, slope, ws, bs, gs, be)

    def timed(fa, fb, n=20):
        """fa on stream a, fb on stream b (either may be None), n iterations each back to back; per-iteration ms of each and the span"""
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        sa.wait_event(start)
        sb.wait_event(start)
        if fa:
            with torch.cuda.stream(sa):
                ev[0].record(sa)
                for _ in range(n):
                    fa()
                ev[1].record(sa)
        if fb:
            with torch.cuda.stream(sb):
                ev[2].record(sb)
                for _ in range(n):
                    fb()
                ev[3].record(sb)
        torch.cuda.synchronize()
        ta = ev[0].elapsed_time(ev[1]) * 1e3 / n if fa else 0.0
        tb = ev[2].elapsed_time(ev[3]) * 1e3 / n if fb else 0.0
        return ta, tb

    def locked(n=20):
        """the two-sided lock of srf_forward_dual with torch events: each stream runs pair, pyramid, pair, ...; a pair launch waits for
        the other stream's latest pair launch"""
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        sa.wait_event(e0)
        sb.wait_event(e0)
        last_b = None
        for k in range(n):
            with torch.cuda.stream(sa):
                if last_b is not None:
                    sa.wait_event(last_b)
                pair()
                ea = torch.cuda.Event()
                ea.record(sa)
                pyr()
            with torch.cuda.stream(sb):
                sb.wait_event(ea)
                pair()
                last_b = torch.cuda.Event()
                last_b.record(sb)
                pyr()
        e1.record(sa)
        e2.record(sb)
        torch.cuda.synchronize()
        return max(e0.elapsed_time(e1), e0.elapsed_time(e2)) * 1e3 / n

    def free(n=20):
        """the same work, the two streams free-running"""
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        sa.wait_event(e0)
        sb.wait_event(e0)
        for s_ in (sa, sb):
            with torch.cuda.stream(s_):
                for k in range(n):
                    pair()
                    pyr()
        e1.record(sa)
        e2.record(sb)
        torch.cuda.synchronize()
        return max(e0.elapsed_time(e1), e0.elapsed_time(e2)) * 1e3 / n

    lk = statistics.median(locked() for _ in range(5))
    fr = statistics.median(free() for _ in range(5))
    print("block period (pair + pyramid on BOTH streams, Bt %d each): phase-locked with events %6.1f us, free-running %6.1f us" % (Bt, lk, fr), flush=True)

    for flags, name in ((0, "pair at 2 blocks / CU"),):
        ops.set_debug_flags(flags)
        rows = []
        for _ in range(5):
            a_alone = timed(pair, None)[0]
            b_alone = timed(None, pyr)[1]
            a_co, b_co = timed(pair, pyr)
            rows.append((a_alone, b_alone, a_co, b_co))
        ops.set_debug_flags(0)
        m = [statistics.median(r[i] for r in rows) for i in range(4)]
        print("%-24s Bt %d per stream: pair alone %6.1f us, pyramid alone %6.1f us | together: pair %6.1f (x %.2f), pyramid %6.1f (x %.2f); "
              "serial %6.1f vs overlapped %6.1f us per (pair + pyramid)" %
              (name, Bt, m[0], m[1], m[2], m[2] / m[0], m[3], m[3] / m[1], m[0] + m[1], max(m[2], m[3])), flush=True)


if __name__ == "__main__":
    main()
