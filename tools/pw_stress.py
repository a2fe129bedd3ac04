#!/usr/bin/env python3
"""Stress the dispatched persistent GEMM instantiations against their pointer forms (bit equality, many repeats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops
DEV = "cuda:0"
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for (Bt, Cin, Cout, L) in [(32, 512, 256, 3200), (32, 256, 512, 3200), (32, 256, 1024, 3200)]:
    x = rnd(Bt, Cin, L, seed=1); w = rnd(Cout, Cin, 1, seed=2, scale=Cin ** -0.5); b = rnd(Cout, seed=3)
    res = rnd(Bt, Cout, L, seed=6)
    sums = torch.zeros(Bt, 64, 2, dtype=torch.float64, device=DEV)
    xf = x.double().reshape(Bt, -1); sums[:, 0, 0] = xf.sum(1); sums[:, 0, 1] = (xf * xf).sum(1)
    for pro in (0, 1, 2, 3):
        kw = {}
        if pro in (1, 2): kw.update(in_sums=sums, in_gamma=rnd(Cin, seed=4) + 1, in_beta=rnd(Cin, seed=5))
        if pro in (2, 3): kw.update(in_prelu=torch.tensor([0.2], device=DEV))
        ops.set_debug_flags(1 << 27); ref = ops.pw_conv(x, w, b, residual=res, **kw)
        nbad = 0
        for _ in range(reps):
            ops.set_debug_flags(0); a = ops.pw_conv(x, w, b, residual=res, **kw)
            nbad += int(not torch.equal(a, ref))
        ops.set_debug_flags(0)
        print((Bt, Cin, Cout, L), "pro", pro, "dispatched != pointer in %d of %d runs" % (nbad, reps))
