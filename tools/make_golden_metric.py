#!/usr/bin/env python3
"""Generate tests/golden/metric_*.npz from the REAL reference validation metric (build container only).

Imports losses/sisdr.py from /root/reference by file path and evaluates PermInvariantSISDR (sisdr.py:66-196) in
fp32 the way the runners construct it (run_improved_sudormrf.py:82-85: zero_mean=True, backward_loss=False,
improvement=True, return_individual_results=True; called with initial_mixtures, :201-205) plus the other flag
combinations, on seeded inputs (oracle/loss_oracle.make_loss_case; mixture = sum of the targets).

    python tools/make_golden_metric.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import load_ref_module  # noqa: E402
from oracle.loss_oracle import make_metric_case  # noqa: E402

# name: (batch, n_src, T, seed, snr_db, mode, zero_mean, improvement, backward_loss, individual)
CASES = {
    "metric_runner_s2": (4, 2, 4000, 11, 5.0, "noisy", True, True, False, True),
    "metric_runner_s3": (3, 3, 1501, 12, 0.0, "noisy", True, True, False, True),
    "metric_runner_cfg_shape": (8, 2, 32000, 13, 12.0, "noisy", True, True, False, True),
    "metric_plain_mean": (5, 2, 777, 14, 3.0, "noisy", False, False, True, False),
    "metric_random_s4": (3, 4, 900, 15, 0.0, "random", True, False, False, True),
    "metric_ragged_lengths": (2, 2, 1000, 16, 8.0, "noisy", True, True, False, True),
    "metric_runner_s5": (2, 5, 800, 17, 2.0, "noisy", True, True, False, True),       # more than 4 sources: the generic kernels
}


def main():
    sisdr = load_ref_module("sudo_rm_rf/dnn/losses/sisdr.py", "_ref_sisdr")
    out_dir = os.path.join(ROOT, "tests", "golden")
    manifest = {}
    for name, (B, S, T, seed, snr, mode, zm, imp, bwd, ind) in CASES.items():
        est_np, tgt_np, mix_np = make_metric_case(name, dict(batch=B, n_src=S, T=T, seed=seed, snr_db=snr, mode=mode))
        fn = sisdr.PermInvariantSISDR(batch_size=B, n_sources=S, zero_mean=zm, backward_loss=bwd, improvement=imp,
                                      return_individual_results=ind)
        with torch.no_grad():
            val, perms = fn(torch.tensor(est_np), torch.tensor(tgt_np), initial_mixtures=torch.tensor(mix_np),
                            return_best_permutation=True)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), value=np.asarray(val.numpy(), np.float32),
                            perms=perms.numpy().astype(np.int32))
        manifest[name] = dict(batch=B, n_src=S, T=T, seed=seed, snr_db=snr, mode=mode, zero_mean=zm, improvement=imp,
                              backward_loss=bwd, individual=ind, mean_value=float(np.mean(val.numpy())))
        print(name, manifest[name])
    json.dump(manifest, open(os.path.join(out_dir, "METRIC_MANIFEST.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
