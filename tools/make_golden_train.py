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
    # the BASELINE shapes of the training path (gradient parity stopped at D <= 4, L <= 240):
    #   cfg 2 (U16 / N512 / D5), batch 2, 1 s @ 8 kHz;  cfg 4 (U36 / N2048 / D6: the CH = 32 SAVE pyramid, K = 2048
    #   weight gradients), batch 1, 0.8 s
    "train_cfg2_shape": (ModelConfig("improved", 256, 512, 16, 5, 21, 512, 2), 2, 8000, 124, 214),   # (seeds with an unclamped loss)
    "train_cfg4_shape": (ModelConfig("improved", 512, 512, 36, 6, 21, 2048, 2), 1, 6400, 105, 205),
    "train_cfg3_shape": (ModelConfig("groupcomm", 256, 512, 8, 5, 21, 512, 2, 1, 16), 2, 8000, 106, 206),
    # ... and at the BENCH length: T = 32000 = 4 s @ 8 kHz, L = 3200 frames -- the tile paths, the
    # full-length SAVE pyramid and the split-K weight gradients that `bench.py --train` times -- batch 4

    "train_cfg2_bench": (ModelConfig("improved", 256, 512, 16, 5, 21, 512, 2), 4, 32000, 124, 214),
    "train_cfg4_bench": (ModelConfig("improved", 512, 512, 36, 6, 21, 2048, 2), 4, 32000, 105, 205),
}
SAMPLE = 4096      # gradient entries kept per parameter (strided)
BIG_SAMPLE = 384   # ... for the BASELINE-shape cases (hundreds of tensors)


def make_batch(cfg, batch, T, seed):
    """(mixture [B,1,T] normalised like the runner, targets [B,S,T])"""
    _, tgt = make_loss_case(batch, cfg.num_sources, T, seed, 5.0, "random")
    tgt = torch.from_numpy(tgt)
    mix = tgt.sum(1, keepdim=True)
    mix = (mix - mix.mean(-1, keepdim=True)) / (mix.std(-1, keepdim=True) + 1e-8)
    return mix, tgt


def sample(g, n=SAMPLE):
    flat = g.reshape(-1)
    step = max(1, flat.size // n)
    return flat[::step][:n].copy(), step


def main():
    ref_imp = load_ref_module("sudo_rm_rf/dnn/models/improved_sudormrf.py", "_ref_improved_sudormrf")
    ref_gc = load_ref_module("sudo_rm_rf/dnn/models/groupcomm_sudormrf_v2.py", "_ref_groupcomm_sudormrf_v2")
    ref_mc = load_ref_module("sudo_rm_rf/dnn/experiments/utils/mixture_consistency.py", "_ref_mixture_consistency")
    sisdr = load_ref_module("sudo_rm_rf/dnn/losses/sisdr.py", "_ref_sisdr")
    loss_fn = sisdr.PITLossWrapper(sisdr.PairwiseNegSDR("sisdr"), pit_from="pw_mtx")
    outdir = os.path.join(ROOT, "tests", "golden")
    manifest = {}
    only = None
    if "--only" in sys.argv:      # (re)generate just these cases, keep the committed others
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
        manifest = json.load(open(os.path.join(outdir, "TRAIN_MANIFEST.json")))
    for name, (cfg, batch, T, wseed, dseed) in CASES.items():
        if only and name not in only:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = (ref_imp.SuDORMRF if cfg.variant == "improved" else ref_gc.GroupCommSudoRmRf)(**cfg.ctor_kwargs())
        sd = make_state_dict(cfg, wseed)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.train()
        mix, tgt = make_batch(cfg, batch, T, dseed)
        # BASELINE-shape cases: the unmodified reference module run in DOUBLE precision; its pad helper builds a float32
        # buffer whatever the input (improved_sudormrf.py:312), so it is bypassed on the instance -- T is a multiple of
        # n_least_samples_req, for which it is the identity (SURVEY.md §8c "fp64 oracle recipe").  At these sizes the
        # reference's own fp32 backward is too noisy to referee anything: its PReLU-slope gradients (one scalar = a sum
        # over ~1e6 terms) differ from the fp64 value by up to 1.2e-2 relative.
        f64 = name.endswith(("_shape", "_bench"))
        if f64:
            assert T % cfg.n_least_samples_req == 0
            model = model.double()
            model.pad_to_appropriate_length = lambda x: x
            mix, tgt = mix.double(), tgt.double()
        # the small cases also pin the gradient w.r.t. the INPUT waveform (round 6: srf_backward_wav), which the reference's
        # autograd returns for a mixture that requires grad -- through the model and, for GroupComm, the mixture-consistency term
        want_gwav = not f64
        if want_gwav:
            mix.requires_grad_()
        rec = model(mix)
        assert rec.dtype == (torch.float64 if f64 else torch.float32)
        if cfg.variant == "groupcomm":
            rec = ref_mc.apply(rec, mix)
        l = torch.clamp(loss_fn(rec, tgt), min=-30.0, max=30.0)
        l.backward()
        arrays = {"loss": np.float32(l.item())}
        if want_gwav:
            arrays["gwav"] = mix.grad.numpy().astype(np.float32)
            mix = mix.detach()
        g32 = None
        if f64:
            # the same step by the reference in its native fp32: how far its OWN gradients are from the fp64 ones, per
            # parameter ("d:" entries) -- the yardstick for what an fp32 implementation can be asked to reproduce
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m32 = (ref_imp.SuDORMRF if cfg.variant == "improved" else ref_gc.GroupCommSudoRmRf)(**cfg.ctor_kwargs())
            m32.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            m32.train()
            rec32 = m32(mix.float())
            if cfg.variant == "groupcomm":
                rec32 = ref_mc.apply(rec32, mix.float())
            torch.clamp(loss_fn(rec32, tgt.float()), min=-30.0, max=30.0).backward()
            g32 = {k: p.grad.numpy().astype(np.float64) for k, p in m32.state_dict(keep_vars=True).items()}
        for k, p in model.state_dict(keep_vars=True).items():
            g = p.grad.numpy()
            if g32 is not None:
                arrays["d:" + k] = np.float64(np.abs(g32[k] - g).max() / max(np.abs(g).max(), 1e-300))
            smp, step = sample(g, BIG_SAMPLE if f64 else SAMPLE)
            arrays["g:" + k] = smp.astype(np.float32)
            arrays["n:" + k] = np.array([step, float(np.abs(g).max()), float(g.astype(np.float64).sum()),
                                         float((g.astype(np.float64) ** 2).sum())])
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **arrays)
        manifest[name] = dict(config=cfg.as_dict(), batch=batch, T=T, weight_seed=wseed, data_seed=dseed,
                              loss=float(l.item()), reference_dtype="float64" if f64 else "float32")
        print(name, manifest[name]["loss"], flush=True)
    json.dump(manifest, open(os.path.join(outdir, "TRAIN_MANIFEST.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
