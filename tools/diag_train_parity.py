#!/usr/bin/env python3
"""Diagnostics for the BASELINE-shape training fixtures: run the step several times, print the worst parameters against
the golden gradients, the run-to-run spread, and the same under kernel-variant flags (bisecting a gradient deviation to a
kernel family).  usage: diag_train_parity.py [case ...]"""
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from test_oracle_golden import train_case  # noqa: E402
from test_gpu_train import build, DEV  # noqa: E402
from sudo_rm_rf_amd import ops  # noqa: E402
import sudo_rm_rf.dnn.experiments.utils.mixture_consistency as mixture_consistency  # noqa: E402
import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib  # noqa: E402


def step(cfg, sd, mix, tgt, flags=0, mode=0):
    model = build(cfg, sd).train()
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    ops.set_debug_flags(flags)
    ops.set_kernel_mode(mode)
    try:
        rec = model(mix.to(DEV))
        if cfg.variant == "groupcomm":
            rec = mixture_consistency.apply(rec, mix.to(DEV))
        l = torch.clamp(loss_fn(rec, tgt.to(DEV)), min=-30., max=+30.)
        l.backward()
        torch.cuda.synchronize()
    finally:
        ops.set_debug_flags(0)
        ops.set_kernel_mode(0)
    return l.item(), {k: p.grad.cpu().numpy().astype(np.float64) for k, p in model.state_dict(keep_vars=True).items()}


def errors(grads, z):
    out = []
    for k, g in grads.items():
        stp, gmax, gsum, gsq = z["n:" + k]
        smp = g.reshape(-1)[::int(stp)][:z["g:" + k].shape[0]]
        rel = np.abs(smp - z["g:" + k]).max() / max(gmax, 1e-12)
        nrm = abs(np.sqrt((g ** 2).sum()) - np.sqrt(gsq)) / max(np.sqrt(gsq), 1e-12)
        dev = float(z["d:" + k]) if "d:" + k in z else 0.0
        out.append((max(rel, nrm), rel, nrm, dev, k))
    return sorted(out, reverse=True)


for name in (sys.argv[1:] or ["train_cfg2_shape", "train_cfg4_shape"]):
    cfg, sd, mix, tgt, z = train_case(name)
    print("=====", name, "loss golden", float(z["loss"]))
    base = None
    for tag, flags, mode in (("default#1", 0, 0), ("default#2", 0, 0), ("default#3", 0, 0), ("fast-fwd(1<<28)", 1 << 28, 0),
                             ("no-rowwise(1<<29)", 1 << 29, 0), ("no-fused-bwd(1<<30)", 1 << 30, 0),
                             ("per-level pyramid(16)", 16, 0), ("mode2 exact MFMA", 0, 2), ("mode1 generic", 0, 1)):
        try:
            loss, g = step(cfg, sd, mix, tgt, flags, mode)
        except Exception as e:  # noqa: BLE001
            print("%-24s FAILED %s" % (tag, e))
            continue
        er = errors(g, z)
        if base is None:
            base = g
        spread = max(np.abs(g[k] - base[k]).max() / max(np.abs(base[k]).max(), 1e-30) for k in g)
        print("%-24s loss %.6f  max run-to-run/variant diff vs #1 %.2e; worst:" % (tag, loss, spread))
        for e in er[:4]:
            print("      %-40s err %.2e (sample %.2e, norm %.2e; reference fp32 own dev %.2e)" % (e[4], e[0], e[1], e[2], e[3]))

# optional: full-tensor comparison against fp64 oracle gradients shipped as tools/tmp_cond_<case>.npz
full = os.path.join(ROOT, "tools", "tmp_cond_cfg2.npz")
if os.path.exists(full) and (not sys.argv[1:] or "train_cfg2_shape" in sys.argv[1:]):
    ref = np.load(full)
    cfg, sd, mix, tgt, z = train_case("train_cfg2_shape")
    for tag, flags in (("default", 0), ("per-level pyramid", 16)):
        _, g = step(cfg, sd, mix, tgt, flags, 0)
        print("== full-tensor check (%s)" % tag)
        rows = []
        for k in g:
            d = np.abs(g[k] - ref[k]); s = max(np.abs(ref[k]).max(), 1e-30)
            rows.append((d.max() / s, k, int(d.argmax()), g[k].shape))
        rows.sort(reverse=True)
        for r in rows[:8]:
            k = r[1]; idx = np.unravel_index(r[2], g[k].shape)
            print("   %-36s max err/max|g| %.2e at %s got %.6e want %.6e (max|g| %.3e)" % (k, r[0], idx, g[k][idx], ref[k][idx], np.abs(ref[k]).max()))
        k = "sm.15.spp_dw.3.conv.weight"
        d = np.abs(g[k] - ref[k]) / np.abs(ref[k]).max()
        bad = np.argwhere(d > 5e-4)
        print("   %s: %d elements off by > 5e-4 of max; first: %s" % (k, len(bad), bad[:12].tolist()))
