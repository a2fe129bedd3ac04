#!/usr/bin/env python3
"""Experiment: the training forward's GEMM on TWO fp16 parts (the default since round 4: 22 mantissa bits, 3 MFMAs per
product block) against round 3's three bf16 parts (debug flag 16384: 24 bits, 6 MFMAs), the exact-fp32 MFMA kernel and the inference two-part
bf16 kernel: error against fp64 on model-sized GEMMs -- including inputs small enough for the fp16 lo parts to go subnormal --
and time per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"
for (Bt, Cin, Cout, L, pro) in [(16, 256, 512, 3200, 0), (16, 512, 256, 3200, 2), (32, 512, 256, 3200, 1)]:
    for xscale in (1.3, 0.02, 30.0):
        g = torch.Generator(device=DEV).manual_seed(5)
        x = torch.randn(Bt, Cin, L, generator=g, device=DEV) * xscale + 0.2 * xscale
        w = torch.randn(Cout, Cin, 1, generator=g, device=DEV) * Cin ** -0.5
        bias = torch.randn(Cout, generator=g, device=DEV) * 0.2
        kw, xin = {}, x.double()
        if pro in (1, 2):
            gamma = torch.randn(Cin, generator=g, device=DEV) * 0.3 + 1.0
            beta = torch.randn(Cin, generator=g, device=DEV) * 0.3
            sums = ops.new_sums(Bt, DEV)
            sums[:, 0, 0] = xin.sum(dim=(1, 2))
            sums[:, 0, 1] = (xin * xin).sum(dim=(1, 2))
            kw.update(in_sums=sums, in_gamma=gamma, in_beta=beta)
            mean = xin.mean(dim=(1, 2), keepdim=True)
            var = (xin * xin).mean(dim=(1, 2), keepdim=True) - mean * mean
            xin = gamma.double().view(1, -1, 1) * (xin - mean) / torch.sqrt(var + 1e-8) + beta.double().view(1, -1, 1)
        if pro == 2:
            kw.update(in_prelu=torch.tensor([0.17], device=DEV))
            xin = torch.where(xin >= 0, xin, 0.17 * xin)
            kw.update(residual=torch.randn(Bt, Cout, L, generator=g, device=DEV))
        cols = torch.arange(0, L, 17, device=DEV)
        want = torch.einsum("mk,bkl->bml", w[:, :, 0].double(), xin[:, :, cols]) + bias.double().view(1, -1, 1)
        if pro == 2:
            want = want + kw["residual"].double()[:, :, cols]
        scale = float(want.abs().max())
        out = {}

        def run(name, fn, flags=0):
            ops.set_debug_flags(flags)
            y = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ops.set_debug_flags(0)
            out[name] = (float((y[:, :, cols].double() - want).abs().max()), e0.elapsed_time(e1) * 100)

        ops.set_debug_flags(16384)
        p3 = ops.pack3_pw_weight(w)
        ops.set_debug_flags(0)
        run("three bf16 parts", lambda: ops.pw_conv3(x, w, bias, p3, **kw), 16384)
        p4 = ops.pack3_pw_weight(w)
        run("two fp16 parts  ", lambda: ops.pw_conv3(x, w, bias, p4, **kw))
        p2 = ops.pack_pw_weight(w)
        run("two bf16 parts  ", lambda: ops.pw_conv(x, w, bias, packed=p2, **kw))
        ops.set_kernel_mode(2)
        run("exact fp32 MFMA ", lambda: ops.pw_conv(x, w, bias, **kw))
        ops.set_kernel_mode(0)
        print("shape %s pro %d x-scale %g (|y| max %.2f): " % ((Bt, Cin, Cout, L), pro, xscale, scale) +
              " | ".join("%s %.2e %.0f us" % (k, e, t) for k, (e, t) in out.items()), flush=True)
