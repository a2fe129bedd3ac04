#!/usr/bin/env python3
"""Generate tests/golden/loss_*.npz from the REAL reference loss (build container only).

Imports losses/sisdr.py from /root/reference by file path, evaluates
    torch.clamp(PITLossWrapper(PairwiseNegSDR("sisdr"), pit_from='pw_mtx')(est, tgt), -30, 30)
(run_improved_sudormrf.py:63-66,169-171) in fp32 on seeded inputs (oracle/loss_oracle.make_loss_case) and stores
loss, pairwise matrix, best permutation and the autograd gradient w.r.t. the estimates.

    python tools/make_golden_loss.py
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
from oracle.loss_oracle import make_loss_case  # noqa: E402

# name: (batch, n_src, T, seed, snr_db, mode)
CASES = {
    "loss_noisy_s2": (4, 2, 4000, 1, 5.0, "noisy"),
    "loss_noisy_s3": (3, 3, 1501, 2, 0.0, "noisy"),
    "loss_high_snr": (2, 2, 8000, 3, 25.0, "noisy"),
    "loss_random": (5, 2, 777, 4, 0.0, "random"),
    "loss_exact_clamped": (2, 2, 1000, 5, 0.0, "exact"),
    "loss_zero_clamped": (2, 2, 1000, 6, 0.0, "zero"),
    "loss_cfg4_shape": (8, 2, 32000, 7, 8.0, "noisy"),
    # more than 4 sources (VERDICT r4 missing 4: the reference accepts n_src < 10, sisdr.py:275): the generic kernels
    "loss_noisy_s5": (3, 5, 1200, 8, 4.0, "noisy"),
    "loss_random_s7": (2, 7, 640, 9, 0.0, "random"),
}


# the other PairwiseNegSDR configurations (sisdr.py:418-424): name: (case tuple, sdr_type, zero_mean, take_log);
# written by `make_golden_loss.py --variants` (LOSSV_MANIFEST.json), the runner configuration above stays untouched
VARIANTS = {
    "lossv_snr": ((4, 2, 3000, 21, 5.0, "noisy"), "snr", True, True),
    "lossv_sdsdr": ((3, 3, 1501, 22, 3.0, "noisy"), "sdsdr", True, True),
    "lossv_sisdr_raw_mean": ((4, 2, 2000, 23, 8.0, "noisy"), "sisdr", False, True),
    "lossv_sisdr_nolog": ((3, 2, 2500, 24, 6.0, "noisy"), "sisdr", True, False),
    "lossv_snr_nolog_raw_mean": ((2, 3, 999, 25, 2.0, "noisy"), "snr", False, False),
    "lossv_sdsdr_random": ((3, 2, 777, 26, 0.0, "random"), "sdsdr", True, True),
}


def main():
    sisdr = load_ref_module("sudo_rm_rf/dnn/losses/sisdr.py", "_ref_sisdr")
    variants = "--variants" in sys.argv
    out_dir = os.path.join(ROOT, "tests", "golden")
    manifest = {}
    todo = {k: (v, "sisdr", True, True) for k, v in CASES.items()} if not variants else VARIANTS
    for name, ((B, S, T, seed, snr, mode), sdr_type, zm, tl) in todo.items():
        loss_fn = sisdr.PITLossWrapper(sisdr.PairwiseNegSDR(sdr_type, zero_mean=zm, take_log=tl), pit_from="pw_mtx")
        pw_fn = sisdr.PairwiseNegSDR(sdr_type, zero_mean=zm, take_log=tl)
        est_np, tgt_np = make_loss_case(B, S, T, seed, snr, mode)
        est = torch.tensor(est_np, requires_grad=True)
        tgt = torch.tensor(tgt_np)
        raw = loss_fn(est, tgt)
        l = torch.clamp(raw, min=-30.0, max=30.0)
        l.backward()
        with torch.no_grad():
            pw = pw_fn(est, tgt)
            _, idx = loss_fn.find_best_perm(pw, S)
        g = est.grad.numpy()
        keep = min(T, 2000)      # gradient: a prefix per row plus row-wise checksums keeps the files small
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), loss=np.float32(l.item()), raw=np.float32(raw.item()),
                            pw=pw.numpy(), perm_index=idx.numpy().astype(np.int32), grad_prefix=g[..., :keep],
                            grad_sum=g.sum(-1).astype(np.float64), grad_sqsum=(g.astype(np.float64) ** 2).sum(-1))
        manifest[name] = dict(batch=B, n_src=S, T=T, seed=seed, snr_db=snr, mode=mode, loss=float(l.item()),
                              raw=float(raw.item()))
        if variants:
            manifest[name].update(sdr_type=sdr_type, zero_mean=zm, take_log=tl)
        print(name, manifest[name])
    json.dump(manifest, open(os.path.join(out_dir, "LOSSV_MANIFEST.json" if variants else "LOSS_MANIFEST.json"), "w"),
              indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
