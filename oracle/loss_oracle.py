"""CPU restatement of the reference's training loss -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

    PITLossWrapper(PairwiseNegSDR("sisdr"), pit_from='pw_mtx')          run_improved_sudormrf.py:63-66
    l = torch.clamp(loss(rec_sources_wavs, clean_wavs), min=-30., max=+30.)   run_improved_sudormrf.py:169-171

PairwiseNegSDR.forward            losses/sisdr.py:426-458
PITLossWrapper.forward            losses/sisdr.py:254-311 (pw_mtx branch: :276-278, :299-306)
PITLossWrapper.find_best_perm     losses/sisdr.py:342-387

Pinned against the reference module itself: tools/make_golden_loss.py imports losses/sisdr.py from
/root/reference, evaluates loss / pairwise matrix / best permutation / autograd gradient on seeded inputs and
stores them under tests/golden/loss_*.npz; tests/test_oracle_golden.py checks this file against those.
"""
import itertools

import numpy as np
import torch

EPS = 1e-8


def make_loss_case(batch, n_src, T, seed, snr_db=5.0, mode="noisy"):
    """Seeded (estimates, targets), both [batch, n_src, T] float32.
    mode: 'noisy'  est = permuted, rescaled targets + noise at snr_db (a realistic training state);
          'exact'  est = targets (loss far below the -30 clamp);
          'zero'   est = 0 (loss = +80 dB, above the +30 clamp);
          'random' est independent of the targets."""
    rng = np.random.default_rng(7919 * (seed + 1) + 13)
    tgt = rng.standard_normal((batch, n_src, T)) * rng.uniform(0.3, 2.0, size=(batch, n_src, 1)) \
        + rng.uniform(-0.2, 0.2, size=(batch, n_src, 1))
    if mode == "exact":
        est = tgt.copy()
    elif mode == "zero":
        est = np.zeros_like(tgt)
    elif mode == "random":
        est = rng.standard_normal((batch, n_src, T))
    else:
        est = np.empty_like(tgt)
        for b in range(batch):
            perm = rng.permutation(n_src)
            for i in range(n_src):
                s = tgt[b, perm[i]]
                noise = rng.standard_normal(T) * np.sqrt(np.mean(s ** 2) / 10 ** (snr_db / 10))
                est[b, i] = rng.uniform(0.5, 1.5) * s + noise + rng.uniform(-0.1, 0.1)
    return est.astype(np.float32), tgt.astype(np.float32)


def pairwise_neg_sdr(est, tgt, sdr_type="sisdr", zero_mean=True, take_log=True):
    """PairwiseNegSDR(sdr_type, zero_mean, take_log).forward  (losses/sisdr.py:426-458).
    torch tensors [batch, n_src, T] -> [batch, n_src(est), n_src(tgt)]; differentiable."""
    assert sdr_type in ("snr", "sisdr", "sdsdr")                                # :421
    if zero_mean:                                                               # :431-435
        tgt = tgt - tgt.mean(dim=2, keepdim=True)
        est = est - est.mean(dim=2, keepdim=True)
    s_target = tgt.unsqueeze(1)                                                 # :437  [B,1,S,T]
    s_estimate = est.unsqueeze(2)                                               # :438  [B,S,1,T]
    if sdr_type in ("sisdr", "sdsdr"):                                          # :440-447
        dot = (s_estimate * s_target).sum(dim=3, keepdim=True)
        energy = (s_target ** 2).sum(dim=3, keepdim=True) + EPS
        proj = dot * s_target / energy
    else:                                                                       # :448-450
        proj = s_target.repeat(1, s_target.shape[2], 1, 1)
    noise = s_estimate - s_target if sdr_type in ("sdsdr", "snr") else s_estimate - proj    # :451-454
    sdr = (proj ** 2).sum(dim=3) / ((noise ** 2).sum(dim=3) + EPS)              # :456-457
    if take_log:
        sdr = 10.0 * torch.log10(sdr + EPS)                                     # :458-459
    return -sdr


def pairwise_neg_sisdr(est, tgt):
    """PairwiseNegSDR('sisdr', zero_mean=True, take_log=True).forward -- the runners' configuration."""
    return pairwise_neg_sdr(est, tgt)


def best_perm(pw):
    """find_best_perm (losses/sisdr.py:342-387, perm_reduce=None): pw [B, est, tgt] ->
    (min_loss [B], index into itertools.permutations order, perms [P, S]).  perms[p][j] is the ESTIMATE
    matched with TARGET j (:368 transposes so that dim 1 = sources, dim 2 = estimates)."""
    n_src = pw.shape[1]
    perms = torch.tensor(list(itertools.permutations(range(n_src))), dtype=torch.long)   # :370-371
    pwl = pw.transpose(-1, -2)                                                            # :368
    loss_set = torch.stack([pwl[:, torch.arange(n_src), p].mean(dim=1) for p in perms], dim=1)  # :375-380
    idx = torch.argmin(loss_set, dim=1)                                                   # :385
    return loss_set.gather(1, idx[:, None])[:, 0], idx, perms


def pit_sisdr_loss(est, tgt, clamp=30.0, sdr_type="sisdr", zero_mean=True, take_log=True):
    """The scalar the runner back-propagates: clamp(mean_b min_perm mean_j pw[b, perm_j, j], -30, 30)."""
    pw = pairwise_neg_sdr(est, tgt, sdr_type, zero_mean, take_log)
    min_loss, idx, perms = best_perm(pw)
    raw = min_loss.mean()                                                       # sisdr.py:307
    return (torch.clamp(raw, min=-clamp, max=clamp) if clamp else raw), raw, pw, perms[idx]


def loss_and_grad(est_np, tgt_np, clamp=30.0, dtype=torch.float64, sdr_type="sisdr", zero_mean=True, take_log=True):
    """numpy in / numpy out: (clamped loss, raw loss, pw [B,S,S], matched estimate per target [B,S],
    d clamped_loss / d est [B,S,T])."""
    est = torch.tensor(est_np, dtype=dtype, requires_grad=True)
    tgt = torch.tensor(tgt_np, dtype=dtype)
    l, raw, pw, match = pit_sisdr_loss(est, tgt, clamp, sdr_type, zero_mean, take_log)
    l.backward()
    return (float(l.detach()), float(raw.detach()), pw.detach().numpy(), match.numpy().astype(np.int32), est.grad.numpy())


def perm_invariant_sisdr(pr, tgt, mix=None, zero_mean=False, improvement=False, backward_loss=True,
                         return_individual_results=False, eps=1e-9):
    """PermInvariantSISDR(n_sources=S, zero_mean, backward_loss, improvement, return_individual_results)
    .forward(pr, tgt, eps, initial_mixtures=mix, return_best_permutation=True)   (losses/sisdr.py:66-196).
    numpy arrays [batch, n_src, T] (mix [batch, 1, T]) -> (value, best permutation index per example); fp64."""
    pr, tgt = np.asarray(pr, np.float64), np.asarray(tgt, np.float64)
    min_len = min(pr.shape[-1], tgt.shape[-1])                                   # normalize_input :97-113
    if mix is not None:
        mix = np.asarray(mix, np.float64)
        min_len = min(min_len, mix.shape[-1])
        mix = mix[:, :, :min_len]
    pr, tgt = pr[:, :, :min_len], tgt[:, :, :min_len]
    if zero_mean:
        pr = pr - pr.mean(-1, keepdims=True)
        tgt = tgt - tgt.mean(-1, keepdims=True)
        if mix is not None:
            mix = mix - mix.mean(-1, keepdims=True)
    S = pr.shape[1]

    def dot(x, y):
        return (x * y).sum(-1, keepdims=True)

    tt = dot(tgt, tgt)                                                           # :136

    def sisnrs(p):                                                               # compute_permuted_sisnrs :119-128
        s_t = dot(p, tgt) / (tt + eps) * tgt
        e_t = p - s_t
        return 10 * np.log10(dot(s_t, s_t) / (dot(e_t, e_t) + eps))

    allp = np.concatenate([sisnrs(pr[:, list(perm), :]) for perm in itertools.permutations(range(S))], -1)   # :138-145
    mean_over_src = allp.mean(-2)                                                # [B, P]
    best_idx = mean_over_src.argmax(-1)                                          # torch.max: first maximum
    best = mean_over_src.max(-1)
    if improvement:                                                              # :149-154
        base = sisnrs(np.repeat(mix, S, axis=1))
        best = best - base.mean()
    if not return_individual_results:
        best = best.mean()
    return (-best if backward_loss else best), best_idx


def make_metric_case(name, c):
    """Inputs of a tests/golden/metric_*.npz case (tools/make_golden_metric.py): estimates, targets, mixture = sum
    of the targets; the 'ragged' case hands in signals of three different lengths."""
    est, tgt = make_loss_case(c["batch"], c["n_src"], c["T"], c["seed"], c["snr_db"], c["mode"])
    mix = tgt.sum(1, keepdims=True).astype(np.float32)
    if name == "metric_ragged_lengths":
        est, mix = est[..., :c["T"] - 37], mix[..., :c["T"] - 5]
    return est, tgt, mix
