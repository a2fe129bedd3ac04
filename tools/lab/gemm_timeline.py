#!/usr/bin/env python3
"""In-kernel timeline of the round-3 256 x 128 GEMM (srf_pwconv_x3w.hip, debug flag 1 << 25): per k-step of a tile, the shader
cycles every wavefront spends in the step's counted wait (vmcnt / lgkmcnt) and in its barrier; per tile, s_memrealtime
(100 MHz, chip-wide) at tile start / k-loop end / epilogue end.  Prints the per-step averages and how the tiles of
different blocks line up in time.  The trace buffer travels in the otherwise unused `mul` pointer."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
SHAPES = {"proj_1x1": (32, 256, 512, 3200, 0, False), "res_conv": (32, 512, 256, 3200, 2, True)}


def main():
    lib = _lib.load()
    out = {}
    for name, (Bt, Cin, Cout, L, pro, res) in SHAPES.items():
        g = torch.Generator(device=DEV).manual_seed(0)
        x = torch.randn(Bt, Cin, L, generator=g, device=DEV)
        w = torch.randn(Cout, Cin, 1, generator=g, device=DEV) * Cin ** -0.5
        bias = torch.randn(Cout, generator=g, device=DEV)
        y = torch.empty(Bt, Cout, L, device=DEV)
        packed = ops.pack_pw_weight(w)
        norm = None
        keep = []
        if pro == 2:
            sums = ops.gln_stats(x, Bt)
            gamma, beta = torch.rand(Cin, generator=g, device=DEV) + 0.5, torch.randn(Cin, generator=g, device=DEV)
            slope = torch.tensor([0.25], device=DEV)
            keep = [sums, gamma, beta, slope]
            norm = ops._norm(sums, gamma, beta, slope)
        resid = torch.randn(Bt, Cout, L, generator=g, device=DEV) if res else None
        osums = None if res else ops.new_sums(Bt, DEV)
        nblk = 256
        trace = torch.zeros(nblk * 8 * 192, dtype=torch.int32, device=DEV)

        def run(flags):
            ops.set_debug_flags(flags)
            rc = lib.srf_pw_conv_packed(_lib.ptr(x), _lib.ptr(w), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(y), Bt, Cin, Cout, L,
                                        norm, _lib.ptr(resid), _lib.ptr(osums), 0, trace.data_ptr(), 1, _lib.current_stream(DEV))
            _lib.check(rc, "srf_pw_conv_packed")

        us = {}
        tl_flags = (1 << 25) | (int(os.environ.get("TL_EXTRA", "0")))
        for flags in (0, tl_flags):
            for _ in range(3):
                run(flags)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(flags)
            e1.record()
            torch.cuda.synchronize()
            us[flags] = e0.elapsed_time(e1) * 100
        ops.set_debug_flags(0)
        t = trace.cpu().numpy().view(np.uint32).reshape(nblk, 8, 3, 64)
        nk = Cin // 32
        wait, bar, ab = t[:, :, 0, :nk].astype(np.float64), t[:, :, 1, :nk].astype(np.float64), t[:, :, 2, :60].astype(np.int64)
        kcyc = t[:, :, 0, 32:52].astype(np.float64)               # [blk, wave, tile]: k-loop shader cycles
        ab = ab.reshape(nblk, 8, 20, 3)
        ntile = (ab[:, 0, :, 0] != 0).sum(axis=1)                 # tiles per block
        print("== %s: %.1f us plain, %.1f us instrumented; tiles per block: %s" % (name, us[0], us[tl_flags], np.bincount(ntile)))
        per_tile_w = wait.sum(axis=(0, 1)) / (ntile.sum() * 8.0)
        per_tile_b = bar.sum(axis=(0, 1)) / (ntile.sum() * 8.0)
        print("  cycles per (wave, tile) by k-step:  wait  " + " ".join("%5.0f" % v for v in per_tile_w))
        print("                                      barrier " + " ".join("%5.0f" % v for v in per_tile_b))
        conv = t[:, :, 1, 32:32 + nk].astype(np.float64).sum(axis=(0, 1)) / (ntile.sum() * 8.0)
        print("                                      convert (incl. its wait for the loads, + ~100 of stamping) " + " ".join("%5.0f" % v for v in conv))
        t0 = ab[:, :, 0, 0].min()
        rel = (ab - t0) / 100.0                                   # us
        kdur, edur = [], []
        for b in range(nblk):
            for i in range(ntile[b]):
                kdur.append(rel[b, 0, i, 1] - rel[b, 0, i, 0])
                edur.append(rel[b, 0, i, 2] - rel[b, 0, i, 1])
        print("  k-loop %.2f us / tile (p10 %.2f p90 %.2f), epilogue %.2f us / tile (p10 %.2f p90 %.2f)"
              % (np.mean(kdur), np.percentile(kdur, 10), np.percentile(kdur, 90), np.mean(edur), np.percentile(edur, 10),
                 np.percentile(edur, 90)))
        for i in range(int(ntile.max())):
            has = ntile > i
            st, ke, ee = rel[has, 0, i, 0], rel[has, 0, i, 1], rel[has, 0, i, 2]
            mhz = (kcyc[has, 0, i] / np.maximum(ke - st, 1e-3)).mean()
            print("  tile %d (%3d blocks): start %6.1f +- %4.1f  k-loop end %6.1f +- %4.1f  epilogue end %6.1f +- %4.1f us;  k-loop %6.0f cycles = %4.0f MHz"
                  % (i, has.sum(), st.mean(), st.std(), ke.mean(), ke.std(), ee.mean(), ee.std(), kcyc[has, 0, i].mean(), mhz))
        out[name] = {"us_plain": us[0], "us_instrumented": us[tl_flags], "wait_cycles_by_kstep": per_tile_w.tolist(),
                     "barrier_cycles_by_kstep": per_tile_b.tolist(), "kloop_us": float(np.mean(kdur)), "epilogue_us": float(np.mean(edur))}
        del keep
    print(json.dumps(out))


if __name__ == "__main__":
    main()
