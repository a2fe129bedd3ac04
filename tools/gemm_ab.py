#!/usr/bin/env python3
"""Same-process interleaved A/B of the 256 x 128 split-bf16 GEMM variants on the model's launch shapes (cfg 2, cfg 4, cfg 5):
rounds of N launches per variant, variants interleaved, median / min us per launch, outputs compared bit for bit.

    python tools/gemm_ab.py [name=debug_flags ...]    default: x3w=0 x3p=8192 (debug flag 8192 on a stand-alone call = the paired-block form of the 256 x 128 kernel)
    GEMM_SHAPES=res_conv,proj_1x1 GEMM_ROUNDS=7 GEMM_ITERS=20"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = {  # name: (Bt, Cin, Cout, L, prologue, epilogue)
    "proj_1x1": (32, 256, 512, 3200, 0, "sums"),
    "res_conv": (32, 512, 256, 3200, 2, "residual"),
    "bottleneck": (32, 512, 256, 3200, 1, "sums"),
    "mask": (32, 256, 1024, 3200, 3, "mask"),
    "cfg4_proj": (32, 512, 512, 3200, 0, "sums"),
    "cfg4_res_conv": (32, 512, 512, 3200, 2, "residual"),
    "cfg4_bottleneck": (32, 2048, 512, 3200, 1, "sums"),
    "cfg5_res_conv": (16, 512, 512, 12800, 2, "residual"),
    "cfg5_mask": (16, 512, 8192, 12800, 3, "mask"),
    # tile counts that are whole multiples of 256 and 512 (no leftover round: steady-state comparison of kernel forms)
    "proj_4096": (32, 256, 512, 4096, 0, "sums"),
    "res_conv_4096": (32, 512, 256, 4096, 2, "residual"),
}


def main():
    variants = [a.split("=") for a in sys.argv[1:]] or [["x3w", "0"], ["x3p", "8192"]]
    variants = [(n, f if ":" in f else f + ":0") for n, f in variants]

    class _Flags:   # debug flags of one variant
        @staticmethod
        def set(spec):
            ops.set_debug_flags(int(spec.split(":")[0]))

    only = os.environ.get("GEMM_SHAPES")
    rounds, iters = int(os.environ.get("GEMM_ROUNDS", "5")), int(os.environ.get("GEMM_ITERS", "10"))
    out = {}
    for name, (Bt, Cin, Cout, L, pro, epi) in SHAPES.items():
        if only and name not in only.split(","):
            continue
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
        ref, times = None, {n: [] for n, _ in variants}
        same = {}
        ref_sums = None
        for n, f in variants:
            _Flags.set(f)
            if "out_sums" in kw:
                kw["out_sums"] = ops.new_sums(Bt, DEV)
            y = ops.pw_conv(x, w, bias, **kw)
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            same[n] = bool(torch.equal(y, ref))
            if "out_sums" in kw:        # the statistics epilogue: bucket totals against the first variant's (fp32 partial sums: ~1e-6)
                tot = kw["out_sums"].sum(dim=1)
                if ref_sums is None:
                    ref_sums = tot
                same[n] = same[n] and bool(torch.allclose(tot, ref_sums, rtol=1e-5, atol=1e-3))
            del y
        for _ in range(rounds):
            for n, f in variants:
                _Flags.set(f)
                ops.pw_conv(x, w, bias, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops.pw_conv(x, w, bias, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) * 1e3 / iters)
        _Flags.set("0:0")
        flop = 2.0 * Bt * Cin * Cout * L
        for n, _ in variants:
            med, mn = statistics.median(times[n]), min(times[n])
            out["%s/%s" % (name, n)] = {"us_median": round(med, 1), "us_min": round(mn, 1),
                                        "TF_fp32_equiv": round(flop / med / 1e6, 1), "bit_equal_to_first": same[n]}
            print("%-16s %-10s median %8.1f us  min %8.1f us  %6.1f TF  bit-equal %s" % (name, n, med, mn, flop / med / 1e6, same[n]),
                  flush=True)
        del x, w, kw, ref
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
