#!/usr/bin/env python3
"""In-kernel timeline of the split-bf16 GEMM (diagnostics): every wavefront stamps s_memtime at fixed
points of each k-tile (library debug flag 9<<16, buffer handed over through the otherwise unused `mul`
pointer).  Prints the average time a wavefront spends between consecutive stamps."""
import json
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def main():
    Bt, Cin, Cout, L = 32, 256, 512, 3200
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(Bt, Cin, L, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, generator=g) * Cin ** -0.5).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    y = torch.empty(Bt, Cout, L, device=DEV)
    nblk = Bt * ((Cout + 127) // 128) * ((L + 127) // 128)
    trace = torch.zeros(nblk * 8 * 64, dtype=torch.int32, device=DEV)
    sums = ops.new_sums(Bt, DEV)
    lib = _lib.load()

    def run(flags):
        ops.set_debug_flags(flags)
        rc = lib.srf_pw_conv_packed(_lib.ptr(x), _lib.ptr(w), None, _lib.ptr(bias), _lib.ptr(y), Bt, Cin, Cout, L,
                                    None, None, _lib.ptr(sums), 0, trace.data_ptr(), 1, _lib.current_stream(DEV))
        _lib.check(rc, "srf_pw_conv")

    for flags in (0, 9 << 16):
        for _ in range(3):
            run(flags)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(flags)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print("flags %#x: %.1f us per launch" % (flags, us))
    ops.set_debug_flags(0)
    t = trace.cpu().numpy().view(np.uint32).reshape(nblk, 8, 64)
    nk = Cin // 32
    NS = 7   # stamps per k-tile
    nst = 3 + NS * nk - 1 + 2   # start, prologue split, pre-loop, NS per k-tile (last one has no store), post-loop, end
    ts = t[:, :, :nst].astype(np.int64)
    hw, xcc = t[:, :, 62], t[:, :, 63]
    d = np.diff(ts, axis=2) & 0xFFFFFFFF                     # [blk, wave, nst-1]
    dur = (ts[:, :, -1] - ts[:, :, 0]) & 0xFFFFFFFF
    t0 = ts[0, 0, 0]
    rel = (ts - t0 + (1 << 31)) % (1 << 32) - (1 << 31)      # signed offsets, wrap-safe
    span = rel.max() - rel.min()
    print("ticks: kernel span %d, mean wave duration %.0f (min %d max %d)" % (span, dur.mean(), dur.min(), dur.max()))
    m = d.mean(axis=(0, 1))
    out = {"span_ticks": int(span), "mean_wave_ticks": float(dur.mean()), "us_per_launch": us, "phases": {}}
    rt = (t[:, :, 61].astype(np.int64) - t[:, :, 60].astype(np.int64)) & 0xFFFFFFFF
    mhz = dur / np.maximum(rt, 1) * 100.0
    print("shader clock seen by the wavefronts (s_memtime / s_memrealtime): mean %.0f MHz (p5 %.0f, p95 %.0f)"
          % (mhz.mean(), np.percentile(mhz, 5), np.percentile(mhz, 95)))
    out["shader_mhz"] = float(mhz.mean())
    print("  prologue: first operands landed + split %6.0f ; store + 3rd gload + barrier %6.0f ; loop entry %4.0f"
          % (m[0], m[1], m[2]))
    # stamps per k-tile: top, operands-landed, split-done, stores-done, gloads-issued, ks0-reads-landed, mma-issued
    names = ["wait-operands", "split VALU", "lds-store", "gload issue", "lds-read ks0", "mfma ks0 + read/mfma ks1",
             "barrier"]
    full = d[:, :, 3:3 + NS * (nk - 1)].reshape(nblk, 8, nk - 1, NS).mean(axis=(0, 1))
    for k in range(nk - 1):
        print("  k%d: " % k + "  ".join("%s %5.0f" % (n.split()[0], v) for n, v in zip(names, full[k])))
    tot = full.mean(axis=0)
    for n, v in zip(names, tot):
        print("  per k-tile  %-40s %7.0f ticks  (%4.1f %%)" % (n, v, 100 * v / tot.sum()))
        out["phases"][n] = float(v)
    print("  per k-tile  sum %.0f ticks (each stamp itself costs ~%d); last k-tile + tail: %s"
          % (tot.sum(), 70, [int(v) for v in m[3 + NS * (nk - 1):]]))
    out["prologue"] = [float(m[0]), float(m[1])]
    out["tail"] = [float(v) for v in m[3 + NS * (nk - 1):]]
    # per-XCD concurrency: the s_memtime bases differ per XCD
    xid = xcc[:, 0] & 15
    for x0 in range(8):
        sel = xid == x0
        if not sel.any():
            continue
        st_ = ts[sel, 0, 0]; en_ = ts[sel, 0, nst - 1]
        base = st_[0]
        st_ = (st_ - base + (1 << 31)) % (1 << 32) - (1 << 31)
        en_ = (en_ - base + (1 << 31)) % (1 << 32) - (1 << 31)
        sp = en_.max() - st_.min()
        if x0 < 2:
            print("  XCD %d: %d blocks, span %d ticks (%.0f MHz if the launch took %.1f us), mean block %.0f ticks, "
                  "avg concurrency %.1f blocks" % (x0, sel.sum(), sp, sp / us, us, (en_ - st_).mean(),
                                                    (en_ - st_).sum() / sp))
    # residency: blocks per CU over time (start order), distinct (xcc, se, cu)
    cu = ((xcc[:, 0].astype(np.int64) & 15) << 16) | (hw[:, 0].astype(np.int64) & 0xFF00) | ((hw[:, 0] >> 13) & 7)
    print("distinct CUs seen: %d; blocks per CU: mean %.1f" % (len(np.unique(cu)), nblk / len(np.unique(cu))))
    st = rel[:, 0, 0] - rel.min()
    en = rel[:, 0, nst - 1] - rel.min()
    order = np.argsort(st)
    print("block start ticks (sorted) every 400th:", st[order][::400].tolist())
    print("block duration by start order every 400th:", (en - st)[order][::400].tolist())
    os.makedirs("gpurun_out/timeline", exist_ok=True)
    json.dump(out, open("gpurun_out/timeline/gemm_timeline.json", "w"), indent=1)
    np.save("gpurun_out/timeline/trace.npy", t[:512])


if __name__ == "__main__":
    main()
