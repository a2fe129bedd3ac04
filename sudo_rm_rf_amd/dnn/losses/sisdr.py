"""Host-side mirror of the part of the reference's ``sudo_rm_rf.dnn.losses.sisdr`` that its runners train with
(SURVEY.md §8 a19):

    PITLossWrapper(PairwiseNegSDR("sisdr"), pit_from='pw_mtx')      experiments/run_improved_sudormrf.py:63-66
    PairwiseNegSDR                                                  losses/sisdr.py:390-458
    PITLossWrapper                                                  losses/sisdr.py:199-387
    PermInvariantSISDR (the runners' validation metric, SI-SDRi)    losses/sisdr.py:66-196; run_improved_sudormrf.py:82-85

Same class names, constructor arguments and call signatures; the arithmetic runs in csrc/srf_loss.hip (one
streaming pass for the forward, one for the gradient).  PairwiseNegSDR takes all of its configurations
("sisdr" | "sdsdr" | "snr", zero_mean, take_log); ``pit_from`` other than 'pw_mtx' or a custom ``perm_reduce`` (which
need loss functions this file of the reference does not contain) raise NotImplementedError instead of falling back to
a CPU/ATen path.  Up to 9 sources, the reference's own limit (sisdr.py:275).
"""
import ctypes as C
import itertools

import torch
from torch import nn
from torch.nn.modules.loss import _Loss

from ... import _lib


def _check(est, tgt):
    if est.shape != tgt.shape:                      # sisdr.py:427
        raise AssertionError("targets and estimates must have the same size, got %s vs %s" %
                             (tuple(tgt.shape), tuple(est.shape)))
    if est.dim() != 3:
        raise RuntimeError("expected [batch, n_src, time], got %s" % (tuple(est.shape),))
    if est.device.type != "cuda" or tgt.device != est.device:
        raise _lib.SrfError("sudo_rm_rf_amd losses run on an MI355X only (estimates on %s, targets on %s); there "
                            "is deliberately no CPU fallback" % (est.device, tgt.device))
    # the reference's own limit (PITLossWrapper.forward, sisdr.py:275): up to 9 sources.  1..4 run on the streaming kernels with
    # everything in registers, 5..9 on the generic forms of csrc/srf_loss.hip (S! permutations per example, as in the reference)
    assert est.shape[1] < 10, f"Expected source axis along dim 1, found {est.shape[1]}"


_SDR_TYPES = {"sisdr": 0, "sdsdr": 1, "snr": 2}
_DEFAULT_VARIANT = (0, 1, 1)      # PairwiseNegSDR("sisdr", zero_mean=True, take_log=True)


def _forward(est, tgt, want_pw, variant=_DEFAULT_VARIANT):
    lib = _lib.load()
    Bt, S, T = est.shape
    dev = est.device
    work = torch.empty(lib.srf_pit_sisdr_work_bytes(Bt, S), dtype=torch.uint8, device=dev)
    loss = torch.empty(2, dtype=torch.float32, device=dev)
    pw = torch.empty((Bt, S, S), dtype=torch.float32, device=dev) if want_pw else None
    rc = lib.srf_pit_sdr_forward(_lib.ptr(est), _lib.ptr(tgt), Bt, S, T, C.c_float(0.0), variant[0], variant[1],
                                 variant[2], _lib.ptr(work), _lib.ptr(pw), _lib.ptr(loss), _lib.current_stream(dev))
    _lib.check(rc, "srf_pit_sdr_forward")
    return work, loss, pw


class _PitSisdr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, tgt, variant=_DEFAULT_VARIANT):
        est_c = est.detach().to(torch.float32).contiguous()
        tgt_c = tgt.detach().to(torch.float32).contiguous()
        with torch.cuda.device(est.device):
            work, loss, _ = _forward(est_c, tgt_c, False, variant)
        ctx.save_for_backward(est_c, tgt_c, work, loss)
        ctx.in_dtype = est.dtype
        return loss[1].clone()

    @staticmethod
    def backward(ctx, grad_out):
        est, tgt, work, loss = ctx.saved_tensors
        Bt, S, T = est.shape
        up = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(est)
        with torch.cuda.device(est.device):
            rc = _lib.load().srf_pit_sisdr_backward(_lib.ptr(est), _lib.ptr(tgt), Bt, S, T, C.c_float(0.0),
                                                    _lib.ptr(work), _lib.ptr(loss), _lib.ptr(up), _lib.ptr(grad),
                                                    _lib.current_stream(est.device))
        _lib.check(rc, "srf_pit_sisdr_backward")
        return grad.to(ctx.in_dtype), None, None


class PairwiseNegSDR(_Loss):
    """Pairwise negative SI-SDR on a batch (reference: losses/sisdr.py:390-458): [batch, n_src, time] x 2 ->
    [batch, n_src (estimates), n_src (targets)]."""

    def __init__(self, sdr_type, zero_mean=True, take_log=True):
        super().__init__()
        assert sdr_type in ["snr", "sisdr", "sdsdr"]            # sisdr.py:421
        self.sdr_type = sdr_type
        self.zero_mean = zero_mean
        self.take_log = take_log

    def _variant(self):
        return (_SDR_TYPES[self.sdr_type], 1 if self.zero_mean else 0, 1 if self.take_log else 0)

    def forward(self, est_targets, targets):
        _check(est_targets, targets)
        if torch.is_grad_enabled() and est_targets.requires_grad:
            raise NotImplementedError("gradients flow through PITLossWrapper(PairwiseNegSDR('sisdr'), "
                                      "pit_from='pw_mtx'); the bare pairwise matrix is forward-only")
        est = est_targets.detach().to(torch.float32).contiguous()
        tgt = targets.detach().to(torch.float32).contiguous()
        with torch.cuda.device(est.device):
            return _forward(est, tgt, True, self._variant())[2]


class PITLossWrapper(nn.Module):
    """Permutation-invariant wrapper (reference: losses/sisdr.py:199-387), 'pw_mtx' mode."""

    def __init__(self, loss_func, pit_from='pw_mtx', perm_reduce=None):
        super().__init__()
        self.loss_func = loss_func
        self.pit_from = pit_from
        self.perm_reduce = perm_reduce
        if self.pit_from not in ['pw_mtx', 'pw_pt', 'perm_avg']:          # sisdr.py:249-251
            raise ValueError('Unsupported loss function type for now. Expected'
                             'one of [`pw_mtx`, `pw_pt`, `perm_avg`]')

    def forward(self, est_targets, targets, return_est=False, reduce_kwargs=None, **kwargs):
        n_src = targets.shape[1]
        assert n_src < 10, f"Expected source axis along dim 1, found {n_src}"      # sisdr.py:274
        if self.pit_from != 'pw_mtx' or self.perm_reduce is not None or not isinstance(self.loss_func, PairwiseNegSDR):
            raise NotImplementedError("the HIP path implements PITLossWrapper(PairwiseNegSDR(...), "
                                      "pit_from='pw_mtx') (run_improved_sudormrf.py:63-66)")
        _check(est_targets, targets)
        variant = self.loss_func._variant()
        mean_loss = _PitSisdr.apply(est_targets, targets, variant)
        if not return_est:
            return mean_loss
        return mean_loss, self.reorder_source(est_targets, targets, variant)

    @staticmethod
    def reorder_source(est_targets, targets, variant=_DEFAULT_VARIANT):
        """Estimates re-ordered so that source j is the estimate matched with target j (sisdr.py:309-311)."""
        lib = _lib.load()
        Bt, S, T = est_targets.shape
        est = est_targets.detach().to(torch.float32).contiguous()
        tgt = targets.detach().to(torch.float32).contiguous()
        with torch.cuda.device(est.device):
            work, _, _ = _forward(est, tgt, False, variant)
            match = torch.empty((Bt, S), dtype=torch.int32, device=est.device)
            rc = lib.srf_pit_sisdr_match(_lib.ptr(work), Bt, S, _lib.ptr(match), _lib.current_stream(est.device))
            _lib.check(rc, "srf_pit_sisdr_match")
        idx = match.long().unsqueeze(-1).expand(Bt, S, T)
        return torch.gather(est_targets, 1, idx)


class PermInvariantSISDR(nn.Module):
    """The runners' validation metric (reference: losses/sisdr.py:66-196): permutation-invariant SI-SNR of
    reconstructed vs target wavs, optionally as an improvement over the input mixture.  Same constructor and
    ``forward`` arguments; one streaming pass over the signals + a per-example finalize (csrc/srf_loss.hip:
    srf_perm_inv_sisdr) instead of the S! materialised permuted copies.  Evaluation only: it raises under autograd
    (the runners train with PITLossWrapper; their PermInvariantSISDR training line is commented out,
    run_improved_sudormrf.py:68-71)."""

    def __init__(self, batch_size=None, zero_mean=False, n_sources=None, backward_loss=True, improvement=False,
                 return_individual_results=False):
        super().__init__()
        self.bs = batch_size
        self.perform_zero_mean = zero_mean
        self.backward_loss = backward_loss
        self.permutations = list(itertools.permutations(torch.arange(n_sources)))
        self.permutations_tensor = torch.LongTensor(self.permutations)
        self.improvement = improvement
        self.n_sources = n_sources
        self.return_individual_results = return_individual_results

    def forward(self, pr_batch, t_batch, eps=1e-9, initial_mixtures=None, return_best_permutation=False):
        if torch.is_grad_enabled() and (pr_batch.requires_grad or t_batch.requires_grad):
            raise NotImplementedError("PermInvariantSISDR is an evaluation metric on the HIP path (no backward): "
                                      "call it under torch.no_grad(), train with PITLossWrapper")
        if pr_batch.dim() != 3 or t_batch.dim() != 3 or pr_batch.shape[:2] != t_batch.shape[:2]:
            raise RuntimeError("expected [batch, n_src, time] estimates and targets, got %s and %s" %
                               (tuple(pr_batch.shape), tuple(t_batch.shape)))
        if pr_batch.shape[1] != self.n_sources:
            raise RuntimeError("constructed for %s sources, got %d" % (self.n_sources, pr_batch.shape[1]))
        if pr_batch.device.type != "cuda" or t_batch.device != pr_batch.device:
            raise _lib.SrfError("sudo_rm_rf_amd losses run on an MI355X only (estimates on %s, targets on %s); "
                                "there is deliberately no CPU fallback" % (pr_batch.device, t_batch.device))
        if self.n_sources > 9:       # (9! permutations per example is where the generic kernel stops; the reference has no limit but
            raise NotImplementedError("the HIP metric supports up to 9 sources, got %d" % self.n_sources)   # materialises [S!, S] indices)
        if self.improvement and initial_mixtures is None:
            raise AttributeError("improvement=True needs initial_mixtures")      # the reference fails on None.repeat
        # normalize_input (sisdr.py:97-113): crop everything to the shortest signal
        min_len = min(pr_batch.shape[-1], t_batch.shape[-1])
        if initial_mixtures is not None:
            min_len = min(min_len, initial_mixtures.shape[-1])
        dev = pr_batch.device
        pr = pr_batch.detach()[:, :, :min_len].to(torch.float32).contiguous()
        tg = t_batch.detach()[:, :, :min_len].to(torch.float32).contiguous()
        mix = None
        if initial_mixtures is not None and self.improvement:
            mix = initial_mixtures.detach()[:, :1, :min_len].to(device=dev, dtype=torch.float32).contiguous()
        Bt, S, T = pr.shape
        lib = _lib.load()
        with torch.cuda.device(dev):
            work = torch.empty(lib.srf_perm_inv_sisdr_work_bytes(Bt, S), dtype=torch.uint8, device=dev)
            best = torch.empty(Bt, dtype=torch.float32, device=dev)
            perm = torch.empty(Bt, dtype=torch.int32, device=dev)
            base = torch.empty(Bt * S, dtype=torch.float32, device=dev) if mix is not None else None
            rc = lib.srf_perm_inv_sisdr(_lib.ptr(pr), _lib.ptr(tg), _lib.ptr(mix), Bt, S, T,
                                        1 if self.perform_zero_mean else 0, C.c_double(float(eps)), _lib.ptr(work),
                                        _lib.ptr(best), _lib.ptr(perm), _lib.ptr(base), _lib.current_stream(dev))
        _lib.check(rc, "srf_perm_inv_sisdr")
        best_sisdr = best
        if self.improvement:
            best_sisdr = best_sisdr - base.mean()             # one batch-and-source mean, as sisdr.py:154
        if not self.return_individual_results:
            best_sisdr = best_sisdr.mean()
        out = -best_sisdr if self.backward_loss else best_sisdr
        if return_best_permutation:
            return out, self.permutations_tensor[perm.long().cpu()]
        return out
