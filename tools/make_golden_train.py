#!/usr/bin/env python3
"""Generate tests/golden/train_*.npz from the REAL reference (build container only): one training step's
gradients.  Reference model (improved_sudormrf.py / groupcomm_sudormrf_v2.py) in train mode, seeded weights and
inputs from oracle/weights.py, loss = clamp(PITLossWrapper(PairwiseNegSDR("sisdr"))(model(mix)[, mixture
consistency], targets), -30, 30) exactly as the runners compute it (run_improved_sudormrf.py:167-171,
run_sudormrf_gc_v2.py:151-160), loss.backward() in fp32 on the CPU.  Stored: loss and, per parameter, the gradient
(small tensors whole, large ones as a strided sample + checksums).

    python tools/make_golden_train.py
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
from oracle.loss_oracle import make_loss_case  # noqa: E402
from oracle.schema import ModelConfig  # noqa: E402
from oracle.weights import make_state_dict  # noqa: E402

# name: (config, batch, T, weight seed, data seed)
CASES = {
    "train_tiny_improved": (ModelConfig("improved", 16, 32, 2, 3, 21, 24, 2), 3, 517, 101, 201),
    "train_improved_mfma": (ModelConfig("improved", 64, 128, 2, 4, 21, 64, 2), 2, 2400, 102, 202),
    "train_tiny_groupcomm": (ModelConfig("groupcomm", 32, 64, 2, 3, 21, 24, 2, 1, 4), 2, 700, 103, 203),
}
SAMPLE = 4096      # gradient entries kept per parameter (strided)


def make_batch(cfg, batch, T, seed):
    """(mixture [B,1,T] normalised like the runner, targets [B,S,T])"""
    _, tgt = make_loss_case(batch, cfg.num_sources, T, seed, 5.0, "random")
    tgt = torch.from_numpy(tgt)
    mix = tgt.sum(1, keepdim=True)
    mix = (mix - mix.mean(-1, keepdim=True)) / (mix.std(-1, keepdim=True) + 1e-8)
    return mix, tgt


def sample(g):
    flat = g.reshape(-1)
    step = max(1, flat.size // SAMPLE)
    return flat[::step][:SAMPLE].copy(), step


def main():
    ref_imp = load_ref_module("sudo_rm_rf/dnn/models/improved_sudormrf.py", "_ref_improved_sudormrf")
    ref_gc = load_ref_module("sudo_rm_rf/dnn/models/groupcomm_sudormrf_v2.py", "_ref_groupcomm_sudormrf_v2")
    ref_mc = load_ref_module("sudo_rm_rf/dnn/experiments/utils/mixture_consistency.py", "_ref_mixture_consistency")
    sisdr = load_ref_module("sudo_rm_rf/dnn/losses/sisdr.py", "_ref_sisdr")
    loss_fn = sisdr.PITLossWrapper(sisdr.PairwiseNegSDR("sisdr"), pit_from="pw_mtx")
    outdir = os.path.join(ROOT, "tests", "golden")
    manifest = {}
    for name, (cfg, batch, T, wseed, dseed) in CASES.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = (ref_imp.SuDORMRF if cfg.variant == "improved" else ref_gc.GroupCommSudoRmRf)(**cfg.ctor_kwargs())
        sd = make_state_dict(cfg, wseed)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.train()
        mix, tgt = make_batch(cfg, batch, T, dseed)
        rec = model(mix)
        if cfg.variant == "groupcomm":
            rec = ref_mc.apply(rec, mix)
        l = torch.clamp(loss_fn(rec, tgt), min=-30.0, max=30.0)
        l.backward()
        arrays = {"loss": np.float32(l.item())}
        for k, p in model.state_dict(keep_vars=True).items():
            g = p.grad.numpy()
            smp, step = sample(g)
            arrays["g:" + k] = smp
            arrays["n:" + k] = np.array([step, float(np.abs(g).max()), float(g.astype(np.float64).sum()),
                                         float((g.astype(np.float64) ** 2).sum())])
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **arrays)
        manifest[name] = dict(config=cfg.as_dict(), batch=batch, T=T, weight_seed=wseed, data_seed=dseed,
                              loss=float(l.item()))
        print(name, manifest[name]["loss"], flush=True)
    json.dump(manifest, open(os.path.join(outdir, "TRAIN_MANIFEST.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
