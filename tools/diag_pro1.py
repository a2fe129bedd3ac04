#!/usr/bin/env python3
"""Diagnostics for the buffer-load instantiation of the persistent GEMM with the GlobLN-only prologue
(srf_pw_bf16x3_p8_kernel<1, true>, dispatched only when SRF_PRO1_BUF is set): where do its results differ from
the pointer form?  Prints the error pattern by example / M tile / time tile / column-in-tile and a channel probe
(one-hot gamma / beta) that attributes the difference to k rows."""
import os, sys, json
os.environ["SRF_PRO1_BUF"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops
DEV = "cuda:0"


def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)


def run(x, w, b, flags, **kw):
    ops.set_debug_flags(flags)
    y = ops.pw_conv(x, w, b, **kw)
    ops.set_debug_flags(0)
    return y


out = {}
Bt, Cin, Cout, L = 32, 512, 256, 3200
x = rnd(Bt, Cin, L, seed=1)
w = rnd(Cout, Cin, 1, seed=2, scale=Cin ** -0.5)
b = rnd(Cout, seed=3)
sums = torch.zeros(Bt, 64, 2, dtype=torch.float64, device=DEV)
xf = x.double().reshape(Bt, -1)
sums[:, 0, 0] = xf.sum(1)
sums[:, 0, 1] = (xf * xf).sum(1)
gamma, beta = rnd(Cin, seed=4) + 1, rnd(Cin, seed=5)
kw = dict(in_sums=sums, in_gamma=gamma, in_beta=beta)
ref = run(x, w, b, 1 << 27, **kw)          # pointer form
mean = (xf.sum(1) / xf.shape[1]).float().view(Bt, 1, 1)
var = ((xf * xf).sum(1) / xf.shape[1]).float().view(Bt, 1, 1) - mean * mean
xn = (x - mean) * torch.rsqrt(var + 1e-8) * gamma.view(1, -1, 1) + beta.view(1, -1, 1)
ref64 = (torch.einsum("mk,bkl->bml", w[:, :, 0].double(), xn.double()) + b.double().view(1, -1, 1)).float()
print("pointer form vs fp64: %.3e" % (ref - ref64).abs().max().item())
for rep in range(3):
    y = run(x, w, b, 0, **kw)
    d = (y - ref).abs()
    print("rep %d: BUF vs pointer max %.3e   BUF vs fp64 %.3e  nonzero diffs %d of %d" %
          (rep, d.max().item(), (y - ref64).abs().max().item(), int((d > 0).sum()), d.numel()))
d = (y - ref).abs()
bad = d > 1e-5
out["bad_fraction"] = float(bad.float().mean())
print("bad (>1e-5) fraction %.4f" % out["bad_fraction"])
print("by example   :", [int(v) for v in bad.sum(dim=(1, 2)).tolist()])
print("by M tile(128):", [int(v) for v in bad.view(Bt, 2, 128, L).sum(dim=(0, 2, 3)).tolist()])
print("by row%32    :", [int(v) for v in bad.view(Bt, 8, 32, L).sum(dim=(0, 1, 3)).tolist()])
print("by time tile :", [int(v) for v in bad.view(Bt, Cout, 25, 128).sum(dim=(0, 1, 3)).tolist()])
print("by col%128/8 :", [int(v) for v in bad.view(Bt, Cout, 25, 16, 8).sum(dim=(0, 1, 2, 4)).tolist()])
# which tiles: first 40 bad (example, mtile, ltile)
tiles = bad.view(Bt, 2, 128, 25, 128).sum(dim=(2, 4))
nz = tiles.nonzero().tolist()
print("bad tiles: %d of %d; first:" % (len(nz), tiles.numel()), nz[:40])
out["bad_tiles"] = len(nz)

# channel probe: x = 0 except channel k0 (constant 1), identity norm -> y[m] = w[m,k0] + bias.  Which k0 go wrong?
x0 = torch.zeros(Bt, Cin, L, device=DEV)
s1 = torch.zeros(Bt, 64, 2, dtype=torch.float64, device=DEV)
s1[:, 0, 1] = Cin * L                     # mean 0, var 1
g1, b0 = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
wrong_k = []
for k0 in range(0, Cin):
    x0.zero_()
    x0[:, k0, :] = 1.0
    a = run(x0, w, b, 1 << 27, in_sums=s1, in_gamma=g1, in_beta=b0)
    c = run(x0, w, b, 0, in_sums=s1, in_gamma=g1, in_beta=b0)
    e = (a - c).abs().max().item()
    if e > 1e-6:
        wrong_k.append((k0, e))
print("x one-hot probe, wrong k0:", wrong_k[:64], "count", len(wrong_k))
# gamma probe: x = 1 everywhere, gamma one-hot at k0, beta 0 -> y[m] = w[m,k0]
x0.fill_(1.0)
s2 = torch.zeros(Bt, 64, 2, dtype=torch.float64, device=DEV)   # sum 0, sumsq = n -> mean 0 var 1 (x is not 0-mean, fine)
s2[:, 0, 1] = Cin * L
wrong_g = []
for k0 in range(0, Cin, 1):
    gk = torch.zeros(Cin, device=DEV)
    gk[k0] = 1.0
    a = run(x0, w, b, 1 << 27, in_sums=s2, in_gamma=gk, in_beta=b0)
    c = run(x0, w, b, 0, in_sums=s2, in_gamma=gk, in_beta=b0)
    e = (a - c).abs().max().item()
    if e > 1e-6:
        wrong_g.append((k0, e))
print("gamma one-hot probe, wrong k0:", wrong_g[:64], "count", len(wrong_g))
wrong_b = []
for k0 in range(0, Cin, 1):
    bk = torch.zeros(Cin, device=DEV)
    bk[k0] = 1.0
    a = run(x0, w, b, 1 << 27, in_sums=s2, in_gamma=b0, in_beta=bk)
    c = run(x0, w, b, 0, in_sums=s2, in_gamma=b0, in_beta=bk)
    e = (a - c).abs().max().item()
    if e > 1e-6:
        wrong_b.append((k0, e))
print("beta one-hot probe, wrong k0:", wrong_b[:64], "count", len(wrong_b))
# statistics probe: does the BUF kernel use the right example's mean/rstd?  x = const per example, gamma 1, beta 0
xs = torch.ones(Bt, Cin, L, device=DEV)
s3 = torch.zeros(Bt, 64, 2, dtype=torch.float64, device=DEV)
for i in range(Bt):       # mean_i = i, var = 1  ->  xn = 1 - i
    n = Cin * L
    s3[i, 0, 0] = i * n
    s3[i, 0, 1] = (1.0 + i * i) * n
a = run(xs, w, b, 1 << 27, in_sums=s3, in_gamma=g1, in_beta=b0)
c = run(xs, w, b, 0, in_sums=s3, in_gamma=g1, in_beta=b0)
print("stats probe by example:", ["%.2e" % v for v in (a - c).abs().amax(dim=(1, 2)).tolist()])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/diag_pro1.json", "w"))
