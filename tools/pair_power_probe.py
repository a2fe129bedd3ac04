#!/usr/bin/env python3
"""Is the fused conv pair power-bound?  Same launch on random operands and on all-zero operands (no bit toggles in the matrix
pipe / on the data buses: same instruction stream, same cycles, far less switching power): time and shader clock
(s_memtime ticks per s_memrealtime microsecond, from the kernel's DBG 2 timeline) of both.

    python tools/pair_power_probe.py [Bt]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def run(Bt, zero):
    lib = _lib.load()
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
    buf = torch.zeros(4096 * 4 * 16, dtype=torch.int32, device=DEV)
    lib.srf_diag_pair_timeline(C.c_void_p(buf.data_ptr()))
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    lib.srf_diag_pair_timeline(C.c_void_p(0))
    t = buf.cpu().view(-1, 16).to(torch.float64)
    t = t[t[:, 5] > 0]
    clk = (t[:, 0] / (t[:, 6] / 100.0)).mean().item()
    print("Bt=%d %-7s operands: %.1f us per launch (min of 5 x 20: %.1f), shader clock %.0f MHz, %d cycles per tile" %
          (Bt, "zero" if zero else "random", sorted(ts)[2], min(ts), clk, t[:, 0].mean().item()))


if __name__ == "__main__":
    for Bt in [int(a) for a in sys.argv[1:]] or [32]:
        run(Bt, False)
        run(Bt, True)
        run(Bt, False)
