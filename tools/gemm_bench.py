#!/usr/bin/env python3
"""Micro-benchmark of the pointwise-conv GEMM (srf_pw_conv) on the cfg-2 shapes, per kernel mode."""
import json
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = {  # name: (Bt, Cin, Cout, L, prologue, residual, stats, mask)
    "proj_1x1": (32, 256, 512, 3200, 0, False, True, False),
    "res_conv": (32, 512, 256, 3200, 2, True, False, False),
    "bottleneck": (32, 512, 256, 3200, 1, False, False, False),
    "mask": (32, 256, 1024, 3200, 3, False, False, True),
    "decoder_frames": (32, 1024, 42, 3200, 0, False, False, False),
}


def main():
    modes = sys.argv[1:] or ["0u", "0p", "2"]   # 0u split-bf16 default (persistent), 0p one tile per block, 2 exact fp32 MFMA
    out = {}
    only = os.environ.get("GEMM_SHAPES")
    iters = int(os.environ.get("GEMM_ITERS", "20"))
    for name, (Bt, Cin, Cout, L, pro, res, stats, mask) in SHAPES.items():
        if only and name not in only.split(","):
            continue
        g = torch.Generator(device="cpu").manual_seed(0)
        x = torch.randn(Bt, Cin, L, generator=g).to(DEV)
        w = (torch.randn(Cout, Cin, 1, generator=g) * Cin ** -0.5).to(DEV)
        bias = torch.randn(Cout, generator=g).to(DEV)
        gamma, beta = torch.rand(Cin, generator=g).to(DEV) + 0.5, torch.randn(Cin, generator=g).to(DEV)
        slope = torch.tensor([0.25], device=DEV)
        kw = {}
        if pro in (1, 2):
            kw.update(in_sums=ops.gln_stats(x, Bt), in_gamma=gamma, in_beta=beta)
        if pro in (2, 3):
            kw.update(in_prelu=slope)
        if res:
            kw.update(residual=torch.randn(Bt, Cout, L, generator=g).to(DEV))
        if mask:
            kw.update(mask_mul=torch.randn(Bt, 512, L, generator=g).to(DEV))
        ref = None
        for mode in modes:
            ops.set_kernel_mode(int(mode[0]))
            ops.set_debug_flags({"p": 2048, "h": 256, "e": 10 << 16}.get(mode[-1], 0))   # p: one tile per block, h: no half-tile tail, e: no epilogue stores
            kw["packed"] = ops.pack_pw_weight(w) if mode in ("0", "0n") else torch.zeros(0, device=DEV)
            if kw["packed"] is None or kw["packed"].numel() == 0:
                kw["packed"] = None
                if mode in ("0", "0n"):
                    continue
            if mode[0] == "0" and len(mode) == 2:
                import sudo_rm_rf_amd.ops as _o
                _pack = _o.pack_pw_weight
                _o.pack_pw_weight = lambda w_: None
            if stats:
                kw["out_sums"] = ops.new_sums(Bt, DEV)
            y = ops.pw_conv(x, w, bias, **kw)
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            err = (y - ref).abs().max().item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = iters
            e0.record()
            for _ in range(n):
                ops.pw_conv(x, w, bias, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            tf = 2.0 * Bt * Cin * Cout * L / (us * 1e-6) / 1e12
            if mode[0] == "0" and len(mode) == 2:
                _o.pack_pw_weight = _pack
            out[f"{name}/mode{mode}"] = {"us": round(us, 1), "TFLOPs_fp32_equiv": round(tf, 1),
                                         "max_abs_diff_vs_first_mode": err}
            print(f"{name:16s} mode {mode:3s}: {us:9.1f} us  {tf:7.1f} TF  diff {err:.2e}", flush=True)
        ops.set_kernel_mode(0)
        ops.set_debug_flags(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
