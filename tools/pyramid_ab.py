#!/usr/bin/env python3
"""Same-process interleaved A/B of the fused pyramid (srf_pyramid: pass 1 + finalize + pass 2) under debug-flag values, with the
in-library profiler's per-kernel times (cfg-2 shapes by default).

    python tools/pyramid_ab.py [flags ...]      default: 0 128 131072    (128: non-persistent pass 1; 1 << 17: pass 1 on the old grid)
    PYR_BT=32 PYR_C=512 PYR_L=3200 PYR_D=5"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"


def main():
    flags = [int(a, 0) for a in sys.argv[1:]] or [0, 128, 1 << 17]
    Bt, C, L, D = (int(os.environ.get(k, d)) for k, d in (("PYR_BT", 32), ("PYR_C", 512), ("PYR_L", 3200), ("PYR_D", 5)))
    g = torch.Generator(device=DEV).manual_seed(0)
    y1 = torch.randn(Bt, C, L, generator=g, device=DEV)
    sums = ops.gln_stats(y1, Bt)
    gam, bet = torch.rand(C, generator=g, device=DEV) + 0.5, torch.randn(C, generator=g, device=DEV) * 0.3
    slope = torch.tensor([0.2], device=DEV)
    ws = [torch.randn(C, 1, 5, generator=g, device=DEV) * 0.4 for _ in range(D)]
    bs = [torch.randn(C, generator=g, device=DEV) * 0.1 for _ in range(D)]
    gs = [torch.rand(C, generator=g, device=DEV) + 0.5 for _ in range(D)]
    be = [torch.randn(C, generator=g, device=DEV) * 0.3 for _ in range(D)]
    ref = None
    per = {f: {} for f in flags}
    for rnd in range(6):
        for f in flags:
            ops.set_debug_flags(f)
            osum = ops.new_sums(Bt, DEV)
            with ops.kernel_trace(DEV) as tr:
                for _ in range(5):
                    out = ops.pyramid(y1, sums, gam, bet, slope, ws, bs, gs, be, out_sums=osum)
            ops.set_debug_flags(0)
            if ref is None:
                ref = out.clone()
            err = float((out - ref).abs().max())
            assert err <= 1e-4, (f, err)
            if rnd == 0:
                continue
            acc = {}
            for name, ms in tr.launches:
                acc.setdefault(name, []).append(ms * 1e3)
            for name, v in acc.items():
                per[f].setdefault(name, []).append(sum(v) / 5)
    for f in flags:
        row = {n: round(statistics.median(v), 1) for n, v in per[f].items()}
        print("flags %-8d total %6.1f us per pyramid  %s" % (f, sum(row.values()), row), flush=True)


if __name__ == "__main__":
    main()
