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


MANV = json.load(open(os.path.join(GOLD, "LOSSV_MANIFEST.json")))     # the other PairwiseNegSDR configurations


@pytest.mark.parametrize("name", sorted(MAN) + sorted(MANV))
def test_pit_sisdr_matches_reference_golden(name):
    c = MAN[name] if name in MAN else MANV[name]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    est_np, tgt_np = loss_oracle.make_loss_case(c["batch"], c["n_src"], c["T"], c["seed"], c["snr_db"], c["mode"])
    sisdr_lib, loss_fn = _loss_fn()
    kw = dict(zero_mean=c.get("zero_mean", True), take_log=c.get("take_log", True))
    sdr_type = c.get("sdr_type", "sisdr")
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR(sdr_type, **kw), pit_from='pw_mtx')
    est = torch.tensor(est_np, device=DEV, requires_grad=True)
    tgt = torch.tensor(tgt_np, device=DEV)
    raw = loss_fn(est, tgt)
    l = torch.clamp(raw, min=-30., max=+30.)                               # run_improved_sudormrf.py:169-171
    l.backward()
    assert abs(l.item() - float(z["loss"])) <= 2e-5 * max(1.0, abs(l.item()))
    assert abs(raw.item() - float(z["raw"])) <= 1e-5 * max(1.0, abs(raw.item())) + 1e-4
    pw = sisdr_lib.PairwiseNegSDR(sdr_type, **kw)(est.detach(), tgt).cpu().numpy()
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
                                         (7, 3, 3, "random"),
                                         (4, 5, 4099, "noisy"), (2, 8, 300, "random")])      # > 4 sources: the generic kernels
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
        sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_pt')(e.to(DEV), t.to(DEV))
    with pytest.raises(AssertionError):
        sisdr_lib.PairwiseNegSDR("sdr")                                     # sisdr.py:421
    with pytest.raises(AssertionError, match="source axis"):               # sisdr.py:275: n_src < 10
        loss_fn(torch.randn(1, 10, 50, device=DEV), torch.randn(1, 10, 50, device=DEV))
    l9 = loss_fn(torch.randn(1, 9, 50, device=DEV), torch.randn(1, 9, 50, device=DEV))      # the limit itself runs (9! permutations)
    assert torch.isfinite(l9).item()
    with pytest.raises(ValueError):
        sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='nope')


METRIC = json.load(open(os.path.join(GOLD, "METRIC_MANIFEST.json")))


@pytest.mark.parametrize("name", sorted(METRIC))
def test_perm_invariant_sisdr_matches_reference_golden(name):
    """The runners' validation metric (PermInvariantSISDR, losses/sisdr.py:66-196) through the module mirror against
    the values of the reference class itself (tools/make_golden_metric.py)."""
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    c = METRIC[name]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    est, tgt, mix = loss_oracle.make_metric_case(name, c)
    fn = sisdr_lib.PermInvariantSISDR(batch_size=c["batch"], n_sources=c["n_src"], zero_mean=c["zero_mean"],
                                      backward_loss=c["backward_loss"], improvement=c["improvement"],
                                      return_individual_results=c["individual"])
    with torch.no_grad():
        val, perms = fn(torch.tensor(est, device=DEV), torch.tensor(tgt, device=DEV),
                        initial_mixtures=torch.tensor(mix, device=DEV), return_best_permutation=True)
        only = fn(torch.tensor(est, device=DEV), torch.tensor(tgt, device=DEV), initial_mixtures=torch.tensor(mix, device=DEV))
    got = val.cpu().numpy()
    assert got.shape == z["value"].shape
    assert (np.abs(got - z["value"]) <= 2e-4 + 2e-5 * np.abs(z["value"])).all()
    assert (perms.numpy() == z["perms"]).all()
    assert torch.equal(only, val)


def test_perm_invariant_sisdr_interface_errors():
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    fn = sisdr_lib.PermInvariantSISDR(batch_size=2, n_sources=2, zero_mean=True, backward_loss=False, improvement=True)
    e, t = torch.randn(2, 2, 100), torch.randn(2, 2, 100)
    with pytest.raises(Exception, match="MI355X"):
        fn(e, t, initial_mixtures=t.sum(1, keepdim=True))                   # CPU tensors: no fallback
    with pytest.raises(AttributeError):
        fn(e.to(DEV), t.to(DEV))                                            # improvement without mixtures
    with pytest.raises(NotImplementedError):
        fn(e.to(DEV).requires_grad_(), t.to(DEV), initial_mixtures=t.sum(1, keepdim=True).to(DEV))
    with pytest.raises(RuntimeError):
        fn(torch.randn(2, 3, 100, device=DEV), torch.randn(2, 3, 100, device=DEV),
           initial_mixtures=torch.randn(2, 1, 100, device=DEV))             # constructed for 2 sources
