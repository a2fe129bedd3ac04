#!/usr/bin/env python3
"""Is the fused conv pair power-bound?  Same launch on random operands and on all-zero operands (no bit toggles in the matrix
pipe / on the data buses: same instruction stream, same cycles, far less switching power): time per launch of both.  (The
shader clocks quoted in DESIGN.md -- 1.4-1.9 GHz on random operands, 2.37 GHz on zeros -- came from the instrumented kernel of
commit 3a712fc, tools/lab/README.md; this script needs only the shipped library.)

    python tools/pair_power_probe.py [Bt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"


def run(Bt, zero):
    K1, Cmid, C2, L = 512, 256, 512, 3200
    g = torch.Generator(device=DEV).manual_seed(0)
    mk = (lambda *s: torch.zeros(*s, device=DEV)) if zero else (lambda *s: torch.randn(*s, generator=g, device=DEV))
    x, res = mk(Bt, K1, L), mk(Bt, Cmid, L)
    w1, w2 = mk(Cmid, K1, 1) * K1 ** -0.5, mk(C2, Cmid, 1) * Cmid ** -0.5
    b1, b2 = mk(Cmid), mk(C2)
    slope = torch.tensor([0.17], device=DEV)
    gamma, beta = (mk(K1) * 0 + (0.0 if zero else 1.0)), mk(K1) * 0.3
    sums = ops.gln_stats(torch.randn(Bt, K1, L, generator=g, device=DEV), Bt)       # (statistics of real data either way: finite rstd)
    p1, p2 = ops.pack_pw_weight(w1), ops.pack_pw_weight(w2)
    f = lambda: ops.pw_conv_pair(x, p1, b1, sums, gamma, beta, slope, res, p2, b2, Cmid, C2)   # noqa: E731
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 50)
    print("Bt=%d %-7s operands: %.1f us per launch (median of 5 x 20; min %.1f)" % (Bt, "zero" if zero else "random", sorted(ts)[2], min(ts)))


if __name__ == "__main__":
    for Bt in [int(a) for a in sys.argv[1:]] or [32]:
        run(Bt, False)
        run(Bt, True)
        run(Bt, False)
