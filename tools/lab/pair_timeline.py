#!/usr/bin/env python3
"""Per-phase shader-clock totals of the fused conv pair (srf_pwconv_x3f.hip, DBG 2): where a wavefront's time goes --
conv 1 (k-loop with activation loads), epilogue 1 (store + hand-over), conv 2 (weight DMA only), epilogues 2.

    python tools/pair_timeline.py [Bt ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def main():
    lib = _lib.load()
    for Bt in [int(a) for a in sys.argv[1:]] or [32, 12]:
        K1, Cmid, C2, L = 512, 256, 512, 3200
        g = torch.Generator(device=DEV).manual_seed(0)
        x = torch.randn(Bt, K1, L, generator=g, device=DEV)
        w1 = torch.randn(Cmid, K1, 1, generator=g, device=DEV) * K1 ** -0.5
        w2 = torch.randn(C2, Cmid, 1, generator=g, device=DEV) * Cmid ** -0.5
        b1, b2 = torch.randn(Cmid, generator=g, device=DEV), torch.randn(C2, generator=g, device=DEV)
        res = torch.randn(Bt, Cmid, L, generator=g, device=DEV)
        slope = torch.tensor([0.17], device=DEV)
        gamma, beta = torch.rand(K1, generator=g, device=DEV) + 0.5, torch.randn(K1, generator=g, device=DEV) * 0.3
        sums = ops.gln_stats(x, Bt)
        p1, p2 = ops.pack_pw_weight(w1), ops.pack_pw_weight(w2)
        buf = torch.zeros(4096 * 4 * 16, dtype=torch.int32, device=DEV)
        run = lambda: ops.pw_conv_pair(x, p1, b1, sums, gamma, beta, slope, res, p2, b2, Cmid, C2)   # noqa: E731
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        e1.synchronize()
        plain = e0.elapsed_time(e1) * 100
        lib.srf_diag_pair_timeline(C.c_void_p(buf.data_ptr()))
        run()
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        e1.synchronize()
        lib.srf_diag_pair_timeline(C.c_void_p(0))
        inst = e0.elapsed_time(e1) * 100
        t = buf.cpu().view(-1, 16).to(torch.float64)
        t = t[t[:, 5] > 0]
        tot = t[:, 0].mean().item()
        clk = (t[:, 0] / (t[:, 6] / 100.0)).mean().item()       # s_memtime ticks per us of s_memrealtime (100 MHz)
        print("Bt=%d: %.1f us plain, %.1f us instrumented; %d wavefronts, %.0f ticks per us (wavefront lifetime %.1f us)" %
              (Bt, plain, inst, t.shape[0], clk, (t[:, 6] / 100.0).mean().item()))
        for nt in sorted(set(t[:, 5].tolist())):
            s = t[t[:, 5] == nt]
            m = s.mean(0)
            print("  %d tile(s): %4d wavefronts | kernel %7.0f ticks | per tile: conv1 %6.0f  epi1 %6.0f  conv2 %6.0f  epi2 %6.0f | counted waits %6.0f  barriers %6.0f | other %5.1f %%" %
                  (nt, s.shape[0], m[0], m[1] / nt, m[2] / nt, m[3] / nt, m[4] / nt, m[7] / nt, m[8] / nt,
                   100 * (m[0] - m[1] - m[2] - m[3] - m[4]) / m[0]))
        del tot


if __name__ == "__main__":
    main()
