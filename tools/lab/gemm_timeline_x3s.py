#!/usr/bin/env python3
"""Per-role in-kernel timeline of the role-split 256 x 128 GEMM (srf_pwconv_x3s.hip, SRF_X3S_TL=1): s_memtime ticks each
wavefront spends in barriers / explicit waits / the epilogue, averaged per role (multipliers 0-7, X stagers 8-9, DMA 10-11)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
SHAPES = {"proj_1x1": (32, 256, 512, 3200, 0, False), "res_conv": (32, 512, 256, 3200, 2, True)}
GEMM = os.environ.get("TL_GEMM", "x3s")
os.environ["SRF_GEMM"] = GEMM
lib = _lib.load()
for name, (Bt, Cin, Cout, L, pro, res) in SHAPES.items():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(Bt, Cin, L, generator=g, device=DEV)
    w = torch.randn(Cout, Cin, 1, generator=g, device=DEV) * Cin ** -0.5
    bias = torch.randn(Cout, generator=g, device=DEV)
    y = torch.empty(Bt, Cout, L, device=DEV)
    packed = ops.pack_pw_weight(w)
    norm, keep = None, []
    if pro == 2:
        sums = ops.gln_stats(x, Bt)
        gamma, beta = torch.rand(Cin, generator=g, device=DEV) + 0.5, torch.randn(Cin, generator=g, device=DEV)
        slope = torch.tensor([0.25], device=DEV)
        keep = [sums, gamma, beta, slope]
        norm = ops._norm(sums, gamma, beta, slope)
    resid = torch.randn(Bt, Cout, L, generator=g, device=DEV) if res else None
    osums = None if res else ops.new_sums(Bt, DEV)
    NB, NW = (512, 8) if GEMM == "x3p" else (256, 12)
    trace = torch.zeros(NB * NW * 8, dtype=torch.int32, device=DEV)

    def run():
        rc = lib.srf_pw_conv_packed(_lib.ptr(x), _lib.ptr(w), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(y), Bt, Cin, Cout, L,
                                    norm, _lib.ptr(resid), _lib.ptr(osums), 0, trace.data_ptr(), 1, _lib.current_stream(DEV))
        _lib.check(rc, "srf_pw_conv_packed")

    us = {}
    for tl in ("0", "1"):
        os.environ["SRF_X3S_TL"] = tl
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us[tl] = e0.elapsed_time(e1) * 100
    t = trace.cpu().numpy().view(np.uint32).reshape(NB, NW, 8).astype(np.float64)
    print("== %s: %.1f us plain, %.1f us instrumented" % (name, us["0"], us["1"]))
    roles = ((("multiply", [0, 1, 2, 3, 4, 5, 6, 7]), ("stage X ", [8, 9]), ("DMA     ", [10, 11])) if GEMM == "x3s" else
             (("all     ", [0, 1, 2, 3, 4, 5, 6, 7]),) if GEMM == "x3p" else
             (("multiply", [0, 1, 2, 4, 5, 6]), ("loaders ", [3, 7])))
    for role, sl in roles:
        r = t[:, sl, :].reshape(-1, 8)
        tot = r[:, 0].mean()
        print("  %s total %8.0f ticks | barrier %5.1f %% | wait %5.1f %% | epilogue %5.1f %% | other %5.1f %% | barriers %4.0f | ticks/us %.0f"
              % (role, tot, 100 * r[:, 1].mean() / tot, 100 * r[:, 2].mean() / tot, 100 * r[:, 3].mean() / tot,
                 100 * (tot - r[:, 1].mean() - r[:, 2].mean() - r[:, 3].mean()) / tot, r[:, 4].mean(), tot / us["1"]))
    del keep
