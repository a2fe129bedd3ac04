#!/usr/bin/env python3
"""Does srf_tac (the lanes kernel, two time steps per lane) compute wrong values when foreign kernels are co-resident?
TAC runs R times on stream A while stream B runs a loop of one kind of foreign kernel; every TAC output is compared
bitwise with the serial result.  Prints the mismatch count per co-runner and the pattern of the mismatching elements."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops
DEV = "cuda:0"
flags = int(os.environ.get("SRF_FLAGS", "0"))
ops.set_debug_flags(flags)


def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)


Bt, G, n, L = int(os.environ.get("BT", "20")), 16, 16, 3200
H = 3 * n
x = rnd(Bt, G, n, L, seed=1)
params = [rnd(H, n, seed=2, scale=n ** -0.5), rnd(H, seed=3, scale=0.1), torch.tensor([0.25], device=DEV),
          rnd(H, H, seed=4, scale=H ** -0.5), rnd(H, seed=5, scale=0.1), torch.tensor([0.2], device=DEV),
          rnd(n, 2 * H, seed=6, scale=(2 * H) ** -0.5), rnd(n, seed=7, scale=0.1), torch.tensor([0.3], device=DEV)]
ref = ops.tac(x, params)
ref2 = ops.tac(x, params)
torch.cuda.synchronize()
print("serial determinism:", bool(torch.equal(ref, ref2)))

# foreign kernels
xg = rnd(32, 256, L, seed=11); wg = rnd(512, 256, 1, seed=12, scale=1 / 16); bg = rnd(512, seed=13)
xr = rnd(32, 512, L, seed=14); wr = rnd(256, 512, 1, seed=15, scale=1 / 22); br = rnd(256, seed=16)
x2 = rnd(12, G, n, L, seed=21)
y32 = rnd(12 * G, 32, L, seed=22)
ws = rnd(32, 16, 1, seed=23, scale=0.25); bs = rnd(32, seed=24)
xs = rnd(12 * G, 16, L, seed=25)
big = rnd(64, 512, L, seed=26)
D = 5
pw = [rnd(32, 1, 5, seed=30 + k, scale=0.4) for k in range(D)]
pb = [rnd(32, seed=40 + k, scale=0.1) for k in range(D)]
pg = [rnd(32, seed=50 + k, scale=0.1) + 1 for k in range(D)]
pbe = [rnd(32, seed=60 + k, scale=0.1) for k in range(D)]
psums = ops.gln_stats(y32, 12 * G)
ig, ib, ip = rnd(32, seed=70, scale=0.1) + 1, rnd(32, seed=71, scale=0.1), torch.tensor([0.25], device=DEV)
def with_mode(fn, mode=0, fl=0):
    def run():
        ops.set_kernel_mode(mode); ops.set_debug_flags(fl)
        try:
            return fn()
        finally:
            ops.set_kernel_mode(0); ops.set_debug_flags(flags)
    return run


xg8 = xg[:8].contiguous()
FOREIGN = {
    "none": lambda: None,
    "gemm one-tile w8 BUF (flag 2048)": with_mode(lambda: ops.pw_conv(xg, wg, bg), 0, 2048),
    "gemm one-tile w8 pointer (2048|1<<27)": with_mode(lambda: ops.pw_conv(xg, wg, bg), 0, 2048 | (1 << 27)),
    "gemm p8 pointer (1<<27)": with_mode(lambda: ops.pw_conv(xg, wg, bg), 0, 1 << 27),
    "gemm p8 no stagger (15<<20)": with_mode(lambda: ops.pw_conv(xg, wg, bg), 0, 15 << 20),
    "gemm exact fp32 MFMA (mode 2)": with_mode(lambda: ops.pw_conv(xg, wg, bg), 2, 0),
    "gemm generic VALU (mode 1, batch 8)": with_mode(lambda: ops.pw_conv(xg8, wg, bg), 1, 0),
    "gemm_p8(proj 256->512)": lambda: ops.pw_conv(xg, wg, bg),
    "gemm_p8(res 512->256)": lambda: ops.pw_conv(xr, wr, br),
    "tac(other buffers)": lambda: ops.tac(x2, params),
    "pw_small(16->32)": lambda: ops.pw_conv(xs, ws, bs),
    "pyramid(C=32)": lambda: ops.pyramid(y32, psums, ig, ib, ip, pw, pb, pg, pbe),
    "torch copy": lambda: big.clone(),
    "torch matmul": lambda: torch.matmul(wr[:, :, 0], xr[0]),
}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
R = int(os.environ.get("R", "24"))
for name, fn in FOREIGN.items():
    outs = []
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        for _ in range(3 * R):
            fn()
    with torch.cuda.stream(sa):
        for _ in range(R):
            outs.append(ops.tac(x, params))
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
    print("%-26s TAC runs differing from serial: %d of %d" % (name, len(bad), R), flush=True)
    for i in bad[:3]:
        d = (outs[i] - ref).abs()
        nz = (d > 0)
        idx = nz.nonzero()
        print("    run %d: %d elements differ, max %.3e; by example %s" %
              (i, int(nz.sum()), float(d.max()), nz.sum(dim=(1, 2, 3)).tolist()))
        print("       by group g:", nz.sum(dim=(0, 2, 3)).tolist())
        print("       by channel i:", nz.sum(dim=(0, 1, 3)).tolist())
        cols = nz.sum(dim=(0, 1, 2))
        print("       by l %% 32:", cols.view(-1, 32).sum(0).tolist())
        lpos = sorted(set((idx[:, 3] // 8 * 8).tolist()))
        print("       distinct 8-column groups (wave tiles) hit: %d; first %s" % (len(lpos), lpos[:12]))
        first = idx[:6].tolist()
        print("       first elements (b,g,i,l):", first, [("%.4f" % float(outs[i][tuple(e)]), "%.4f" % float(ref[tuple(e)])) for e in first])
