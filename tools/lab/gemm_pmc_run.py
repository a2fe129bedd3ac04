#!/usr/bin/env python3
"""Workload for counter passes over the 256 x 128 GEMM variants (tools/gpu_gemm_pmc.sh): a few launches of each model form per
variant.  Variants: x3w (round 3), x3s (round 4, SRF_GEMM=x3s).  GEMM_SHAPES / GEMM_VARIANTS select."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402
from gemm_ab import SHAPES  # noqa: E402

DEV = "cuda:0"
only = os.environ.get("GEMM_SHAPES", "proj_1x1,res_conv").split(",")
variants = os.environ.get("GEMM_VARIANTS", "x3w,x3s").split(",")
n = int(os.environ.get("GEMM_ITERS", "6"))
for name in only:
    Bt, Cin, Cout, L, pro, epi = SHAPES[name]
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(Bt, Cin, L, generator=g, device=DEV)
    w = torch.randn(Cout, Cin, 1, generator=g, device=DEV) * Cin ** -0.5
    bias = torch.randn(Cout, generator=g, device=DEV)
    kw = {}
    if pro in (1, 2):
        kw.update(in_sums=ops.gln_stats(x, Bt), in_gamma=torch.rand(Cin, generator=g, device=DEV) + 0.5,
                  in_beta=torch.randn(Cin, generator=g, device=DEV))
    if pro in (2, 3):
        kw.update(in_prelu=torch.tensor([0.25], device=DEV))
    if epi == "residual":
        kw.update(residual=torch.randn(Bt, Cout, L, generator=g, device=DEV))
    elif epi == "mask":
        kw.update(mask_mul=torch.randn(Bt, Cout // 2, L, generator=g, device=DEV))
    else:
        kw.update(out_sums=ops.new_sums(Bt, DEV))
    kw["packed"] = ops.pack_pw_weight(w)
    for v in variants:
        os.environ["SRF_GEMM"] = v if v != "x3w" else ""
        for _ in range(n):
            ops.pw_conv(x, w, bias, **kw)
        torch.cuda.synchronize()
