#!/usr/bin/env python3
"""Ablation timing of the 256x128 GEMM (srf_pwconv_x3v.hip) on the res_conv shape: flags (abl << 16), plus the sched variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops
DEV = "cuda:0"
Bt, Cin, Cout, L = 32, 512, 256, 3200
g = torch.Generator().manual_seed(0)
x = torch.randn(Bt, Cin, L, generator=g).to(DEV)
w = (torch.randn(Cout, Cin, 1, generator=g) * Cin ** -0.5).to(DEV)
bias = torch.randn(Cout, generator=g).to(DEV)
gamma, beta = torch.rand(Cin, generator=g).to(DEV) + 0.5, torch.randn(Cin, generator=g).to(DEV)
slope = torch.tensor([0.25], device=DEV)
res = torch.randn(Bt, Cout, L, generator=g).to(DEV)
kw = dict(in_sums=ops.gln_stats(x, Bt), in_gamma=gamma, in_beta=beta, in_prelu=slope, residual=res, packed=ops.pack_pw_weight(w))
names = {0: "full", 30: "epilogue without its global stores", 1: "no X loads", 2: "no W DMA", 3: "no loads at all", 4: "no MFMA", 8: "no split", 12: "no split, no MFMA", 16: "no epilogue",
         19: "no loads, no epilogue", 31: "nothing (barriers + addressing only)"}
def run(flags, n=30):
    ops.set_debug_flags(flags)
    for _ in range(3): ops.pw_conv(x, w, bias, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.pw_conv(x, w, bias, **kw)
    e1.record(); torch.cuda.synchronize()
    ops.set_debug_flags(0)
    return e0.elapsed_time(e1) * 1e3 / n
for abl, nm in names.items():
    print("%-40s %8.1f us" % (nm, run(abl << 16)), flush=True)

print("%-40s %8.1f us" % ("half-tile tail (flag 256)", run(256)))
kw.pop("residual")
print("%-40s %8.1f us" % ("full, no residual", run(0)))
print("%-40s %8.1f us" % ("no residual, no stores", run(30 << 16)))
