#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (build container only).

Imports the unmodified reference modules from /root/reference by file path
(under private module names, so they cannot collide with this repo's
``sudo_rm_rf`` compatibility package), loads deterministic weights from
oracle/weights.py, runs the reference forward on CPU in fp32 and stores the
outputs.  The GPU box has no /root/reference: tests there regenerate the same
weights / inputs from (config, seed) and compare against these files.

    python tools/make_golden.py            # writes tests/golden/*.npz + MANIFEST.json
"""
import importlib.util
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.schema import ModelConfig, CONFIGS, state_dict_schema  # noqa: E402
from oracle.weights import make_state_dict, make_mixture  # noqa: E402

REF = os.environ.get("SRF_REFERENCE", "/root/reference")


def load_ref_module(rel, name):
    path = os.path.join(REF, rel)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spec.loader.exec_module(mod)
    return mod


# (name, config, batch, T, weight seed, input seed)
CASES = [
    ("tiny_improved", ModelConfig("improved", 16, 32, 2, 3, 21, 24, 2), 2, 517, 1, 1),
    ("tiny_improved_d1", ModelConfig("improved", 8, 16, 1, 1, 21, 8, 3), 2, 90, 2, 2),
    ("tiny_improved_short", ModelConfig("improved", 16, 32, 1, 4, 21, 16, 2), 1, 50, 3, 3),
    ("tiny_groupcomm", ModelConfig("groupcomm", 32, 64, 2, 3, 21, 24, 2, 1, 4), 2, 700, 4, 4),
    ("tiny_groupcomm_a2", ModelConfig("groupcomm", 32, 64, 1, 2, 11, 16, 2, 2, 8), 1, 333, 5, 5),
    ("cfg1_improved_u8", CONFIGS["cfg1_improved_u8"], 1, 32000, 10, 10),
    ("cfg1_improved_u8_pad", CONFIGS["cfg1_improved_u8"], 1, 32079, 10, 11),
    ("cfg2_improved_u16", CONFIGS["cfg2_improved_u16"], 2, 32000, 20, 20),
    ("cfg3_groupcomm_u8", CONFIGS["cfg3_groupcomm_u8"], 1, 32000, 30, 30),
    ("cfg4_improved_u36_n2048", CONFIGS["cfg4_improved_u36_n2048"], 1, 32000, 40, 40),
    ("cfg5_improved_u36_n4096", CONFIGS["cfg5_improved_u36_n4096"], 1, 128000, 50, 50),
    # the reference's own __main__ smoke configurations (shape-only checks there):
    #   improved_sudormrf.py:321-332  U16 / N512, batch 3, T = 32079 (pad path)
    #   groupcomm_sudormrf_v2.py:421-442  D = 7, K = 91, N = 2048, S = 4, 10 s @ 16 kHz
    ("main_improved_b3_pad", CONFIGS["cfg2_improved_u16"], 3, 32079, 60, 60),
    ("main_groupcomm_d7_k91", ModelConfig("groupcomm", 256, 512, 16, 7, 91, 2048, 4, 1, 16), 1, 160000, 61, 61),
]


def main():
    # --only a,b: (re)generate just these cases and merge them into the committed manifest
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    torch.manual_seed(0)
    ref_imp = load_ref_module("sudo_rm_rf/dnn/models/improved_sudormrf.py", "_ref_improved_sudormrf")
    ref_gc = load_ref_module("sudo_rm_rf/dnn/models/groupcomm_sudormrf_v2.py", "_ref_groupcomm_sudormrf_v2")
    ref_mc = load_ref_module("sudo_rm_rf/dnn/experiments/utils/mixture_consistency.py", "_ref_mixture_consistency")
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    manifest = {"generator": "tools/make_golden.py", "torch": torch.__version__,
                "reference": "etzinis/sudo_rm_rf @ /root/reference", "cases": {}}
    if only and os.path.exists(os.path.join(outdir, "MANIFEST.json")):
        manifest["cases"] = json.load(open(os.path.join(outdir, "MANIFEST.json")))["cases"]
    for name, cfg, batch, T, wseed, iseed in CASES:
        if only and name not in only:
            continue
        t0 = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if cfg.variant == "improved":
                model = ref_imp.SuDORMRF(**cfg.ctor_kwargs())
            else:
                model = ref_gc.GroupCommSudoRmRf(**cfg.ctor_kwargs())
        # schema check against the live reference
        ref_sd = model.state_dict()
        schema = state_dict_schema(cfg)
        assert [k for k, _ in schema] == list(ref_sd.keys()), name
        assert all(tuple(ref_sd[k].shape) == s for k, s in schema), name
        sd = make_state_dict(cfg, wseed)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval()
        A = cfg.in_audio_channels if cfg.variant == "groupcomm" else 1
        wav = make_mixture(batch, T, iseed, channels=A)
        with torch.no_grad():
            out = model(torch.from_numpy(wav))
            arrays = {"out": out.numpy().astype(np.float32)}
            if cfg.variant == "groupcomm" and A == 1 and not name.startswith("main_"):
                mc = ref_mc.apply(out, torch.from_numpy(wav))
                arrays["out_mixture_consistency"] = mc.numpy().astype(np.float32)
        np.savez(os.path.join(outdir, name + ".npz"), **arrays)
        manifest["cases"][name] = {
            "config": cfg.as_dict(), "batch": batch, "T": T, "weight_seed": wseed,
            "input_seed": iseed, "out_shape": list(out.shape),
            "out_abs_max": float(out.abs().max()), "out_rms": float(out.pow(2).mean().sqrt()),
            "params": int(sum(p.numel() for p in model.parameters())),
        }
        print(f"{name:28s} out{tuple(out.shape)} absmax={float(out.abs().max()):.4f} "
              f"rms={float(out.pow(2).mean().sqrt()):.4f}  {time.time() - t0:.1f}s", flush=True)
    with open(os.path.join(outdir, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
