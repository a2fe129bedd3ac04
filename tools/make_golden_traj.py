#!/usr/bin/env python3
"""Generate tests/golden/*_traj.npz from the REAL reference (build container only): a 3-step TRAINING TRAJECTORY of the runner's
loop body (run_improved_sudormrf.py:146-177) --

    opt.zero_grad(); rec = model(mix); l = clamp(PITLossWrapper(PairwiseNegSDR("sisdr"))(rec, clean), -30, 30); l.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0); opt.step()            # opt = torch.optim.Adam(lr=1e-3)

on three different seeded batches, with the unmodified reference modules.  Gradients and the optimizer are pinned separately
elsewhere (train_*.npz, test_fused_clip_adam_matches_torch); this pins their COMPOSITION over several steps (VERDICT r3 missing 4).

Stored: the three losses and, per parameter, a strided sample of the FINAL weights.  The reference runs in float64 (the clean
target); the same trajectory in the reference's native float32 gives, per parameter, the yardstick "d:<name>" = relative L2
distance of ITS weight change from the float64 one.  That yardstick matters here: Adam's first steps move every weight by ~lr x
sign(gradient), so an element whose gradient is within rounding noise of zero moves by +-lr in either direction whatever the
implementation -- the fp32 reference itself differs from its fp64 run by a few percent of the update on such tensors.

    python tools/make_golden_traj.py
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import load_ref_module  # noqa: E402
from make_golden_train import make_batch, sample  # noqa: E402
from oracle.schema import ModelConfig  # noqa: E402
from oracle.weights import make_state_dict  # noqa: E402

STEPS = 3
# name: (config, batch, T, weight seed, first data seed)
CASES = {
    "train_improved_mfma_traj": (ModelConfig("improved", 64, 128, 2, 4, 21, 64, 2), 2, 2400, 102, 302),
    "train_cfg2_shape_traj": (ModelConfig("improved", 256, 512, 16, 5, 21, 512, 2), 2, 8000, 124, 314),
}
SAMPLE = 2048


def run(ref_imp, loss_fn, cfg, sd, batches, dtype):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = ref_imp.SuDORMRF(**cfg.ctor_kwargs())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    if dtype == torch.float64:
        model = model.double()
        model.pad_to_appropriate_length = lambda x: x      # (builds a float32 buffer whatever the input, improved_sudormrf.py:312)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for mix, tgt in batches:
        opt.zero_grad()
        rec = model(mix.to(dtype))
        l = torch.clamp(loss_fn(rec, tgt.to(dtype)), min=-30.0, max=30.0)
        l.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        losses.append(float(l.item()))
    return losses, {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}


def main():
    ref_imp = load_ref_module("sudo_rm_rf/dnn/models/improved_sudormrf.py", "_ref_improved_sudormrf")
    sisdr = load_ref_module("sudo_rm_rf/dnn/losses/sisdr.py", "_ref_sisdr")
    loss_fn = sisdr.PITLossWrapper(sisdr.PairwiseNegSDR("sisdr"), pit_from="pw_mtx")
    outdir = os.path.join(ROOT, "tests", "golden")
    mpath = os.path.join(outdir, "TRAIN_MANIFEST.json")
    manifest = json.load(open(mpath))
    for name, (cfg, batch, T, wseed, dseed) in CASES.items():
        assert T % cfg.n_least_samples_req == 0
        sd = make_state_dict(cfg, wseed)
        batches = [make_batch(cfg, batch, T, dseed + s) for s in range(STEPS)]
        l64, w64 = run(ref_imp, loss_fn, cfg, sd, batches, torch.float64)
        l32, w32 = run(ref_imp, loss_fn, cfg, sd, batches, torch.float32)
        arrays = {"losses": np.array(l64), "losses_fp32": np.array(l32)}
        for k, w in w64.items():
            w0 = sd[k].astype(np.float64)
            d64, d32 = w - w0, w32[k] - w0
            arrays["d:" + k] = np.float64(np.sqrt(((d32 - d64) ** 2).sum()) / max(np.sqrt((d64 ** 2).sum()), 1e-300))
            smp, step = sample(w, SAMPLE)
            arrays["w:" + k] = smp
            arrays["n:" + k] = np.array([step, float(np.sqrt((d64 ** 2).sum()))])
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **arrays)
        manifest[name] = dict(config=cfg.as_dict(), batch=batch, T=T, weight_seed=wseed, data_seed=dseed, steps=STEPS,
                              loss=l64[0], losses=l64, reference_dtype="float64", lr=1e-3, clip_grad_norm=5.0)
        print(name, l64, l32, "worst fp32-vs-fp64 update deviation %.3g" % max(float(arrays[k]) for k in arrays if k.startswith("d:")),
              flush=True)
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
