"""GPU parity of the training loss (csrc/srf_loss.hip) through the reference's own interface
(sudo_rm_rf.dnn.losses.sisdr) against oracle/loss_oracle.py and the reference-generated fixtures."""
import itertools
import json
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
MAN = json.load(open(os.path.join(GOLD, "LOSS_MANIFEST.json")))


def _loss_fn():
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib                        # the reference's import path
    return sisdr_lib, sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')


@pytest.mark.parametrize("name", sorted(MAN))
def test_pit_sisdr_matches_reference_golden(name):
    c = MAN[name]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    est_np, tgt_np = loss_oracle.make_loss_case(c["batch"], c["n_src"], c["T"], c["seed"], c["snr_db"], c["mode"])
    sisdr_lib, loss_fn = _loss_fn()
    est = torch.tensor(est_np, device=DEV, requires_grad=True)
    tgt = torch.tensor(tgt_np, device=DEV)
    raw = loss_fn(est, tgt)
    l = torch.clamp(raw, min=-30., max=+30.)                               # run_improved_sudormrf.py:169-171
    l.backward()
    assert abs(l.item() - float(z["loss"])) <= 2e-5 * max(1.0, abs(l.item()))
    assert abs(raw.item() - float(z["raw"])) <= 1e-5 * max(1.0, abs(raw.item())) + 1e-4
    pw = sisdr_lib.PairwiseNegSDR("sisdr")(est.detach(), tgt).cpu().numpy()
    # at -110 dB (estimate == target) the reference's own fp32 round-off is ~1e-4 dB: tolerance relative to |pw|
    assert (np.abs(pw - z["pw"]) <= 1e-4 + 5e-6 * np.abs(z["pw"])).all()
    g = est.grad.cpu().numpy()
    k = z["grad_prefix"].shape[-1]
    scale = max(np.abs(z["grad_prefix"]).max(), 1e-12)
    assert np.abs(g[..., :k] - z["grad_prefix"]).max() <= 2e-5 * scale
    assert np.abs(g.sum(-1) - z["grad_sum"]).max() <= 1e-5
    # re-ordered estimates: source j <- the estimate matched with target j
    perms = list(itertools.permutations(range(c["n_src"])))
    match = np.array([perms[i] for i in z["perm_index"]])
    _, reordered = loss_fn(est.detach(), tgt, return_est=True)
    want = np.take_along_axis(est_np, match[:, :, None], axis=1)
    assert np.array_equal(reordered.cpu().numpy(), want)


@pytest.mark.parametrize("Bt,S,T,mode", [(32, 2, 32000, "noisy"), (3, 4, 515, "noisy"), (1, 1, 64, "random"),
                                         (7, 3, 3, "random")])
def test_pit_sisdr_matches_fp64_oracle(Bt, S, T, mode):
    est_np, tgt_np = loss_oracle.make_loss_case(Bt, S, T, 100 + Bt + S, 3.0, mode)
    _, raw64, pw64, match64, g64 = loss_oracle.loss_and_grad(est_np, tgt_np, clamp=0.0)
    _, loss_fn = _loss_fn()
    est = torch.tensor(est_np, device=DEV, requires_grad=True)
    tgt = torch.tensor(tgt_np, device=DEV)
    (2.5 * loss_fn(est, tgt)).backward()                                   # upstream gradient is a device scalar
    g = est.grad.cpu().numpy()
    assert np.abs(g - 2.5 * g64).max() <= 2e-5 * max(np.abs(g64).max() * 2.5, 1e-12)
    assert abs(loss_fn(est.detach(), tgt).item() - raw64) <= 1e-4


def test_pit_sisdr_interface_errors():
    sisdr_lib, loss_fn = _loss_fn()
    e, t = torch.randn(2, 2, 100), torch.randn(2, 2, 100)
    with pytest.raises(Exception, match="MI355X"):
        loss_fn(e, t)                                                       # CPU tensors: no fallback
    with pytest.raises(AssertionError):
        loss_fn(e.to(DEV), torch.randn(2, 2, 99, device=DEV))
    with pytest.raises(NotImplementedError):
        sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("snr"), pit_from='pw_mtx')(e.to(DEV), t.to(DEV))
    with pytest.raises(ValueError):
        sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='nope')
