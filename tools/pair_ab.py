#!/usr/bin/env python3
"""Same-process interleaved A/B of the fused conv pair (srf_pw_conv_pair, srf_pwconv_x3f.hip) against the two launches it
replaces (res_conv / bottleneck, then proj_1x1; one-block kernel and, with debug flag 8192, the paired-block kernel that
srf_forward used up to round 4): median / min us, outputs compared bit for bit.

    python tools/pair_ab.py [Bt ...]        default 32 20 16 12 (cfg 2's batch and the engine's stream-split sub-batches)
    PAIR_ROUNDS=7 PAIR_ITERS=20 PAIR_SHAPES=res,head"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    batches = [int(a) for a in sys.argv[1:]] or [32, 20, 16, 12]
    rounds, iters = int(os.environ.get("PAIR_ROUNDS", "5")), int(os.environ.get("PAIR_ITERS", "10"))
    shapes = os.environ.get("PAIR_SHAPES", "res,head").split(",")
    L, Cmid, C2 = 3200, 256, 512
    out = {}
    for shape in shapes:
        K1 = 512
        for Bt in batches:
            g = torch.Generator(device=DEV).manual_seed(0)
            x = torch.randn(Bt, K1, L, generator=g, device=DEV) * 1.3 + 0.2
            w1 = torch.randn(Cmid, K1, 1, generator=g, device=DEV) * K1 ** -0.5
            b1 = torch.randn(Cmid, generator=g, device=DEV)
            w2 = torch.randn(C2, Cmid, 1, generator=g, device=DEV) * Cmid ** -0.5
            b2 = torch.randn(C2, generator=g, device=DEV)
            res = torch.randn(Bt, Cmid, L, generator=g, device=DEV) if shape == "res" else None
            slope = torch.tensor([0.17], device=DEV) if shape == "res" else None
            gamma, beta = torch.rand(K1, generator=g, device=DEV) + 0.5, torch.randn(K1, generator=g, device=DEV) * 0.3
            sums = ops.gln_stats(x, Bt)
            p1, p2 = ops.pack_pw_weight(w1), ops.pack_pw_weight(w2)
            osum = ops.new_sums(Bt, DEV)
            if not ops.pw_conv_pair_supported(Bt, K1, Cmid, C2, L):
                print("%s Bt=%d: not served" % (shape, Bt))
                continue

            def two(flags):
                def f():
                    ops.set_debug_flags(flags)
                    y = ops.pw_conv(x, w1, b1, in_sums=sums, in_gamma=gamma, in_beta=beta, in_prelu=slope, residual=res, packed=p1)
                    y2 = ops.pw_conv(y, w2, b2, out_sums=osum, packed=p2)
                    ops.set_debug_flags(0)
                    return y, y2
                return f

            def pair(flags):
                def f():
                    ops.set_debug_flags(flags)
                    r = ops.pw_conv_pair(x, p1, b1, sums, gamma, beta, slope, res, p2, b2, Cmid, C2, out_sums2=osum)
                    ops.set_debug_flags(0)
                    return r
                return f

            variants = {"two_x3w": two(0), "two_x3p": two(8192), "pair": pair(0), "pair_drain": pair(1 << 23)}
            # PAIR_FLAGS="name=flags,...": extra variants of the pair under diagnostic flags (round 6: 1 << 18 the CU's second
            # block at s_setprio 1, 1 << 19 / 1 << 20 that block half a k-step / half a tile late)
            for item in filter(None, os.environ.get("PAIR_FLAGS", "").split(",")):
                nm, fl = item.split("=")
                variants["pair_" + nm] = pair(int(fl, 0))
            ref = variants["two_x3w"]()
            equal = {}
            for name, fn in variants.items():
                y, y2 = fn()
                equal[name] = bool(torch.equal(y, ref[0]) and torch.equal(y2, ref[1]))
            torch.cuda.synchronize()
            times = {n: [] for n in variants}
            for _ in range(rounds):
                for name, fn in variants.items():
                    fn()
                    times[name].append(timed(fn, iters))
            row = {}
            for name in variants:
                row[name] = {"median_us": round(statistics.median(times[name]), 1), "min_us": round(min(times[name]), 1),
                             "bit_equal": equal[name]}
                print("%-5s Bt=%-3d %-11s median %7.1f us  min %7.1f us  bit-equal %s" %
                      (shape, Bt, name, row[name]["median_us"], row[name]["min_us"], equal[name]), flush=True)
            out["%s_bt%d" % (shape, Bt)] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
