"""Whole-model parity through the C ABI (srf_forward) on the GPU.

Bar (BASELINE.json north_star): separated waveforms within 1e-4 max-abs of the reference CPU forward
on the same inputs.  The golden outputs were produced by the unmodified reference (tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_case
from oracle import np_oracle, torch_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4   # north_star tolerance, max-abs

ALL_CASES = ["tiny_improved", "tiny_improved_d1", "tiny_improved_short", "tiny_groupcomm", "tiny_groupcomm_a2",
             "cfg1_improved_u8", "cfg1_improved_u8_pad", "cfg2_improved_u16", "cfg3_groupcomm_u8",
             "cfg4_improved_u36_n2048", "cfg5_improved_u36_n4096",
             "main_improved_b3_pad", "main_groupcomm_d7_k91"]


def build(cfg, sd):
    import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf       # reference import paths
    import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as sudormrf_gc_v2
    cls = improved_sudormrf.SuDORMRF if cfg.variant == "improved" else sudormrf_gc_v2.GroupCommSudoRmRf
    m = cls(**cfg.ctor_kwargs())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(DEV).eval()


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from sudo_rm_rf_amd import ops
    ops.set_kernel_mode(0)


@pytest.mark.parametrize("name", ALL_CASES)
def test_forward_matches_reference_golden(manifest, name):
    cfg, sd, wav, gold = load_case(manifest, name)
    model = build(cfg, sd)
    with torch.no_grad():
        out = model(torch.from_numpy(wav).to(DEV))
    assert out.dtype == torch.float32 and out.device.type == "cuda"
    got = out.cpu().numpy()
    assert got.shape == gold["out"].shape
    err = np.abs(got - gold["out"]).max()
    print(f"{name}: max abs err vs reference = {err:.3e} (out abs max {np.abs(gold['out']).max():.3f})")
    assert err <= TOL
    if "out_mixture_consistency" in gold:
        import sudo_rm_rf.dnn.experiments.utils.mixture_consistency as mixture_consistency
        mc = mixture_consistency.apply(out, torch.from_numpy(wav).to(DEV)).cpu().numpy()
        assert np.abs(mc - gold["out_mixture_consistency"]).max() <= TOL


@pytest.mark.parametrize("name", ["tiny_improved", "tiny_groupcomm", "cfg1_improved_u8_pad"])
def test_generic_kernels_agree_with_fast(manifest, name):
    from sudo_rm_rf_amd import ops
    cfg, sd, wav, gold = load_case(manifest, name)
    model = build(cfg, sd)
    x = torch.from_numpy(wav).to(DEV)
    try:
        ops.set_kernel_mode(1)
        with torch.no_grad():
            gen = model(x).cpu().numpy()
    finally:
        ops.set_kernel_mode(0)
    assert np.abs(gen - gold["out"]).max() <= TOL


@pytest.mark.parametrize("name", ["tiny_improved", "tiny_groupcomm"])
def test_intermediates_match_oracle(manifest, name):
    cfg, sd, wav, _ = load_case(manifest, name)
    model = build(cfg, sd)
    tr = {}
    np_oracle.forward(cfg, sd, wav, dtype=np.float64, trace=tr)
    with torch.no_grad():
        model(torch.from_numpy(wav).to(DEV))
    plan = model._engine().last_plan
    Bt, L = wav.shape[0], plan.frames
    enc = plan.debug_fetch(0, (Bt, cfg.enc_num_basis, L)).cpu().numpy()
    assert np.abs(enc - tr["enc"]).max() < 1e-5
    last = f"sm.{cfg.num_blocks - 1}.out" if cfg.variant == "improved" else f"sm.{cfg.num_blocks - 1}.UBlock.out"
    smo = plan.debug_fetch(1, (Bt, cfg.out_channels, L)).cpu().numpy()
    assert np.abs(smo - tr[last].reshape(smo.shape)).max() < 2e-4
    masked = plan.debug_fetch(2, tr["masked"].shape).cpu().numpy()
    assert np.abs(masked - tr["masked"]).max() < 1e-4


@pytest.mark.parametrize("name", ["cfg1_improved_u8", "cfg3_groupcomm_u8"])
def test_unfused_pyramid_agrees(manifest, name):
    """debug flag 16 = per-level depthwise/merge kernels instead of the fused pyramid."""
    from sudo_rm_rf_amd import ops
    cfg, sd, wav, gold = load_case(manifest, name)
    model = build(cfg, sd)
    try:
        ops.set_debug_flags(16)
        with torch.no_grad():
            out = model(torch.from_numpy(wav).to(DEV)).cpu().numpy()
    finally:
        ops.set_debug_flags(0)
    assert np.abs(out - gold["out"]).max() <= TOL


# (case, bench batch, kernel families the single-stream forward MUST have been dispatched to)
_X3W = {"pw_conv_x3p<0>", "pw_conv_x3p<1>", "pw_conv_x3p<2>", "pw_mask_decode"}      # (x3p: the paired-block 256 x 128 kernel)
# cfg 2 (B = 256): bottleneck / res_conv run fused with the proj_1x1 that follows (round 5, srf_pwconv_x3f.hip); only the last
# block's res_conv is still a launch of its own
_PAIRS = {"pw_pair_x3f<1>", "pw_pair_x3f<2>", "pw_conv_x3p<2>", "pw_mask_decode"}
_BENCH_BATCH = [("cfg2_improved_u16", 32, _PAIRS), ("cfg3_groupcomm_u8", 32, {"pw_conv_x3p<1>", "pw_mask_decode", "pw_conv_small"}),
                ("cfg4_improved_u36_n2048", 32, _X3W), ("cfg5_improved_u36_n4096", 16, _X3W)]


@pytest.mark.parametrize("case,batch,families", _BENCH_BATCH, ids=[c for c, _, _ in _BENCH_BATCH])
def test_bench_batch_examples_match_reference_golden(manifest, case, batch, families):
    """Every BASELINE configuration AT THE BATCH bench.py TIMES IT (cfg 2 / 3 / 4: 32, cfg 5: 16), through the kernels the
    bench times: every example must reproduce the reference's golden output of the same waveform run on its own -- nothing
    on the path mixes examples -- within the north-star tolerance, on the single-stream forward and on
    every batch split the auto-tuner may pick.  The batch-1/2 goldens alone fall below the `tiles >= #CUs` gate of
    srf_pw_conv_packed and never reach the 256 x 128 GEMM, so the test also proves -- through the
    in-library profiler -- that the single-stream forward really ran those kernel families (a future dispatch change
    cannot silently re-route it)."""
    from sudo_rm_rf_amd import ops
    cfg, sd, wav, gold = load_case(manifest, case)
    model = build(cfg, sd)
    nb = wav.shape[0]
    reps = np.concatenate([wav] * (batch // nb), axis=0)    # golden inputs interleaved
    assert reps.shape[0] == batch
    x = torch.from_numpy(reps).to(DEV)
    eng = model._engine()
    assert eng.multi_stream

    def check(out, what):
        out = out.cpu().numpy()
        worst = max(float(np.abs(out[i] - gold["out"][i % nb]).max()) for i in range(batch))
        print(f"{case} batch {batch} ({what}): max abs err vs reference golden = {worst:.3e}")
        assert worst <= TOL, (what, worst)

    try:
        eng.multi_stream = False
        with torch.no_grad(), ops.kernel_trace(DEV) as tr:
            out = model(x)
        check(out, "single stream")
        missing = families - tr.names
        assert not missing, "single-stream forward did not run %s (ran %s)" % (sorted(missing), sorted(tr.names))
        count = {n: sum(1 for k, _ in tr.launches if k == n) for n in tr.names}
        U = cfg.num_blocks
        if cfg.variant == "improved" and cfg.out_channels == 256:
            # B = 256 (cfg 2): bottleneck + proj_1x1 of block 0 and res_conv of block i + proj_1x1 of block i + 1 as fused pairs
            # (srf_pwconv_x3f.hip); the last res_conv and the mask + decoder on the 256 x 128 kernels
            assert (count["pw_pair_x3f<1>"], count["pw_pair_x3f<2>"], count["pw_conv_x3p<2>"], count["pw_mask_decode"]) == \
                (1, U - 1, 1, 1), count
            assert "pw_conv_x3p<0>" not in count and "pw_conv_x3p<1>" not in count, count
        elif cfg.variant == "improved":   # bottleneck, U x proj_1x1, U x res_conv, mask + decoder -- ALL on the 256 x 128 kernel
            assert (count["pw_conv_x3p<1>"], count["pw_conv_x3p<0>"], count["pw_conv_x3p<2>"], count["pw_mask_decode"]) == \
                (1, U, U, 1), count
        else:                             # GroupComm: bottleneck + mask on it, the per-group convs on the thin-shape kernel
            assert (count["pw_conv_x3p<1>"], count["pw_mask_decode"], count["pw_conv_small"]) == (1, 1, 2 * U), count
        # (the fused tail contracts the masked values with the decoder inside the mask GEMM: no GEMM is left on the 128 x 128
        # kernels and the masked tensor is never stored)
        assert sum(v for k, v in count.items() if k.startswith("pw_conv_bf16x3") or k == "pw_conv_mfma") == 0, count
        assert "pw_conv_x3w<3>" not in count and "transpose" not in count, count
        eng.multi_stream = True
        for parts in eng._split_candidates(batch)[1:]:        # the explicit splits: halves, 5 : 3 and 9 : 7
            out = torch.empty_like(out)
            with torch.no_grad():
                params = [p.detach() for p in model.state_dict(keep_vars=True).values()]
                with torch.cuda.device(x.device), eng._run_lock(x.device):
                    eng._forward_split(parts, x, out, eng._param_table(params, x.device))
            check(out, "split %s" % (parts,))
        with torch.no_grad():                                  # and the public path with whatever the auto-tuner picks
            for _ in range(4):
                out = model(x)
        check(out, "auto-tuned %s" % (eng._split_choice.get((x.device.index, batch, x.shape[-1])),))
    finally:
        eng.multi_stream = True


@pytest.mark.parametrize("case,batch", [("cfg2_improved_u16", 32), ("cfg3_groupcomm_u8", 32), ("cfg1_improved_u8", 24)])
@pytest.mark.parametrize("recipe", [False, True])
def test_fused_tail_agrees_with_materialised_masked_tensor(manifest, case, batch, recipe):
    """K5 A/B: the mask GEMM fused with the decoder's contraction (partial frames, no masked tensor) against the same forward
    with debug flag 32768 = mask GEMM -> masked tensor -> frame GEMM -> overlap-add; plain forward and the separate() recipe
    (rescale + mixture consistency folded into the overlap-add).  Run-to-run identical (fixed summation order)."""
    from sudo_rm_rf_amd import ops
    cfg, sd, wav, gold = load_case(manifest, case)
    model = build(cfg, sd)
    nb = wav.shape[0]
    reps = np.concatenate([wav] * ((batch + nb - 1) // nb), axis=0)[:batch]
    x = torch.from_numpy(reps).to(DEV) * (3.0 if recipe else 1.0) + (0.25 if recipe else 0.0)
    eng = model._engine()

    def run():
        with torch.no_grad(), ops.kernel_trace(DEV) as tr:
            out = eng.separate(model, x, True) if recipe else model(x)
        return out.clone(), tr.names

    try:
        eng.multi_stream = False
        fused, names = run()
        assert "pw_mask_decode" in names, sorted(names)
        again, _ = run()
        assert torch.equal(fused, again)
        ops.set_debug_flags(32768)
        plain, names = run()
        assert "pw_mask_decode" not in names and "pw_conv_x3w<3>" in names, sorted(names)
    finally:
        ops.set_debug_flags(0)
        eng.multi_stream = True
    if not recipe:      # the masked tensor does not exist after a fused forward: asking for it must fail, not return partial frames
        plan = eng.last_plan
        with pytest.raises(Exception, match="not materialised"):
            plan.debug_fetch(2, (batch, cfg.num_sources * cfg.enc_num_basis, plan.frames))
    err = float((fused - plain).abs().max())
    print(f"{case} batch {batch} recipe={recipe}: fused vs materialised tail max abs diff = {err:.3e}")
    assert err <= 2e-5 * max(1.0, float(plain.abs().max()))
    if not recipe:
        worst = max(float(np.abs(fused[i].cpu().numpy() - gold["out"][i % nb]).max()) for i in range(batch))
        assert worst <= TOL


def test_run_to_run_determinism(manifest):
    cfg, sd, wav, _ = load_case(manifest, "cfg1_improved_u8")
    model = build(cfg, sd)
    x = torch.from_numpy(wav).to(DEV)
    with torch.no_grad():
        a = model(x).clone()
        b = model(x).clone()
    # fp64 atomics may reorder, which can move a statistic by one fp32 ulp at most
    assert (a - b).abs().max().item() < 1e-6


def test_arbitrary_lengths_and_dtypes(manifest):
    cfg, sd, _, _ = load_case(manifest, "tiny_improved")
    model = build(cfg, sd)
    sdt = torch_oracle.to_torch(sd)
    for T in (1, 7, 50, 160, 161, 1001):
        g = torch.Generator().manual_seed(T)
        wav = torch.randn(2, 1, T, generator=g)
        with torch.no_grad():
            want = torch_oracle.forward(cfg, sdt, wav)
            got = model(wav.to(DEV))
            got64 = model(wav.double().to(DEV))          # reference casts to fp32 in its pad buffer
        assert got.shape == (2, cfg.num_sources, T)
        assert (got.cpu() - want).abs().max().item() <= TOL, T
        assert torch.equal(got, got64)
    with pytest.raises(RuntimeError):
        model(torch.zeros(2, 100, device=DEV))           # 2-D input: error, like the reference
    with pytest.raises(RuntimeError):
        model(torch.zeros(2, 2, 100, device=DEV))        # wrong channel count


def test_variable_length_inference_keeps_plans_and_memory_bounded():
    """Real inference brings a new length with every call: the engine builds one plan per (batch, length),
    does NOT run its split auto-tune for shapes it has not seen repeatedly, keeps at most _MAX_PLANS plans per device, and
    the memory it holds stops growing once the LRU is full; an evicted shape that comes back still gives the same output."""
    from oracle import weights
    from oracle.schema import ModelConfig
    from sudo_rm_rf_amd import engine as engine_mod
    cfg = ModelConfig("improved", 128, 256, 4, 5, 21, 256, 2)
    model = build(cfg, weights.make_state_dict(cfg, seed=9))
    eng = model._engine()
    g = torch.Generator().manual_seed(0)
    lengths = [int(t) for t in torch.randint(24000, 40000, (30,), generator=g)]
    first = {}
    reserved = []
    with torch.no_grad():
        for i, T in enumerate(lengths):
            wav = torch.randn(8, 1, T, generator=g).to(DEV)
            out = model(wav)
            assert out.shape == (8, 2, T) and torch.isfinite(out).all()
            if i < 2:
                first[T] = (wav, out.clone())
            torch.cuda.synchronize()
            reserved.append(torch.cuda.memory_reserved())
            assert len([k for k in eng._plans if k[0] == torch.cuda.current_device()]) <= engine_mod._MAX_PLANS
            assert not eng._split_choice, "the split auto-tune must not run for one-off shapes"
        assert max(reserved[20:]) <= 1.25 * max(reserved[:12]) + (64 << 20), reserved
        for T, (wav, want) in first.items():           # evicted long ago: rebuilt, same result
            assert torch.equal(model(wav), want)


@pytest.mark.parametrize("T", [32079, 20010], ids=["ragged-3216", "ragged-2016"])
def test_fused_pairs_equal_separate_launches_on_ragged_lengths(T):
    """Whole-model check of the fused conv pairs (round 5) away from the bench shape: a B = 256 model at batch 32 and lengths
    whose frame count is no multiple of the pair kernel's 128-column tile (partial last tiles; the reference's zero right-pad,
    improved_sudormrf.py:303-314, as bounds checks) and that end up with different launch sizes (832 / 512 tiles on 512 block
    slots).  A pair's two OUTPUT TENSORS are bit-identical to the two launches it replaces (test_gpu_ops.py); its GlobLN statistics
    -- fp64 buckets of fp32 partial sums, grouped by a different tile shape -- agree to rounding, so the whole forward with and
    without the pairs (debug flag 1) must agree to rounding (measured 4e-6 of the output scale after three blocks; bar 2e-5), single-stream and split; the profiler
    proves the pairs were dispatched."""
    from oracle import weights
    from oracle.schema import ModelConfig
    from sudo_rm_rf_amd import ops
    cfg = ModelConfig("improved", 256, 512, 3, 5, 21, 512, 2)
    model = build(cfg, weights.make_state_dict(cfg, seed=31))
    eng = model._engine()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(32, 1, T, generator=g).to(DEV)
    try:
        eng.multi_stream = False
        with torch.no_grad(), ops.kernel_trace(DEV) as tr:
            fused = model(x)
        assert {"pw_pair_x3f<1>", "pw_pair_x3f<2>"} <= tr.names, tr.names
        ops.set_debug_flags(1)
        with torch.no_grad(), ops.kernel_trace(DEV) as tr0:
            plain = model(x)
        assert not any(n.startswith("pw_pair") for n in tr0.names), tr0.names
        ops.set_debug_flags(0)
        scale = float(plain.abs().max())
        assert torch.isfinite(fused).all() and float((fused - plain).abs().max()) <= 2e-5 * scale, float((fused - plain).abs().max()) / scale
        eng.multi_stream = True
        out = torch.empty_like(fused)
        with torch.no_grad():
            params = [p.detach() for p in model.state_dict(keep_vars=True).values()]
            for parts in eng._split_candidates(32)[1:]:
                with torch.cuda.device(x.device), eng._run_lock(x.device):
                    eng._forward_split(parts, x, out, eng._param_table(params, x.device))
                assert float((out - plain).abs().max()) <= 2e-5 * scale, parts
    finally:
        ops.set_debug_flags(0)
        eng.multi_stream = True


def test_submodule_forwards(manifest):
    """UConvBlock / TAC / GlobLN called stand-alone (as pickled sub-modules may be) match the oracle."""
    cfg, sd, wav, _ = load_case(manifest, "tiny_groupcomm")
    model = build(cfg, sd)
    tr = {}
    np_oracle.forward(cfg, sd, wav, dtype=np.float64, trace=tr)
    x = torch.from_numpy(tr["bottleneck"].astype(np.float32)).to(DEV)
    with torch.no_grad():
        y = model.sm[0](x).cpu().numpy()
    assert np.abs(y - tr["sm.0.UBlock.out"].reshape(y.shape)).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(nIn=8, nOut=12, kSize=3, stride=2, groups=4), dict(nIn=6, nOut=6, kSize=7, stride=1, groups=1),
                                dict(nIn=16, nOut=16, kSize=1, stride=1, groups=1)],
                         ids=["k3-s2-g4", "k7", "pointwise"])
def test_convnormact_any_kernel_size(kw):
    """The reference's ConvNormAct (improved_sudormrf.py:50-73) is an ordinary nn.Conv1d + GlobLN + PReLU and accepts any
    kernel size / stride / groups; round 6: so does the mirror (general srf_conv1d kernel outside the shapes UConvBlock builds).
    Against the same arithmetic in torch fp64 on the CPU."""
    import torch.nn.functional as F
    import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf
    torch.manual_seed(3)
    m = improved_sudormrf.ConvNormAct(**kw)
    with torch.no_grad():
        m.norm.gamma.uniform_(0.5, 1.5)
        m.norm.beta.normal_(0.0, 0.3)
        m.act.weight.fill_(0.17)
    x = torch.randn(3, kw["nIn"], 157)
    c = m.conv
    y = F.conv1d(x.double(), c.weight.double(), c.bias.double(), c.stride, c.padding, c.dilation, c.groups)
    mean = y.mean(dim=(1, 2), keepdim=True)
    var = ((y - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    want = F.prelu(m.norm.gamma.double().view(1, -1, 1) * (y - mean) / torch.sqrt(var + 1e-8) + m.norm.beta.double().view(1, -1, 1),
                   m.act.weight.double())
    with torch.no_grad():
        got = m.to(DEV)(x.to(DEV)).cpu().double()
    assert got.shape == want.shape and (got - want).abs().max().item() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(nIn=8, nOut=8, kSize=5, stride=1, d=3, groups=8), dict(nIn=4, nOut=10, kSize=3, stride=2, d=2, groups=2),
                                dict(nIn=8, nOut=8, kSize=5, stride=2, d=1, groups=8)],
                         ids=["depthwise-dilated", "grouped-k3-d2", "uconv-shape"])
def test_dilatedconvnorm_any_dilation(kw):
    """DilatedConvNorm (improved_sudormrf.py:138-159) with any kernel size / dilation / groups, as the reference's."""
    import torch.nn.functional as F
    import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf
    torch.manual_seed(4)
    m = improved_sudormrf.DilatedConvNorm(**kw)
    with torch.no_grad():
        m.norm.gamma.uniform_(0.5, 1.5)
        m.norm.beta.normal_(0.0, 0.3)
    x = torch.randn(2, kw["nIn"], 240)
    c = m.conv
    y = F.conv1d(x.double(), c.weight.double(), c.bias.double(), c.stride, c.padding, c.dilation, c.groups)
    mean = y.mean(dim=(1, 2), keepdim=True)
    var = ((y - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    want = m.norm.gamma.double().view(1, -1, 1) * (y - mean) / torch.sqrt(var + 1e-8) + m.norm.beta.double().view(1, -1, 1)
    with torch.no_grad():
        got = m.to(DEV)(x.to(DEV)).cpu().double()
    assert got.shape == want.shape and (got - want).abs().max().item() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("K", [11, 21], ids=["generic-encoder", "k21-encoder"])
@pytest.mark.parametrize("variant", ["improved", "groupcomm"])
def test_separate_pipeline_matches_reference_recipe(variant, K):
    """sudo_rm_rf_amd.pipeline.separate (srf_separate: statistics kernel, normalise in the encoder's load, rescale and
    mixture consistency in the overlap-add) == the README's normalise / model / rescale (/ mixture consistency) lines
    evaluated with the oracle model on the CPU (README.md:100-114), and == the three-kernel form around model()."""
    from oracle import weights
    from oracle.schema import ModelConfig
    from sudo_rm_rf_amd import ops, pipeline
    cfg = (ModelConfig("improved", 16, 32, 2, 3, K, 24, 2) if variant == "improved"
           else ModelConfig("groupcomm", 16, 32, 2, 3, K, 24, 2, 1, 4))
    sd = weights.make_state_dict(cfg, seed=5)
    model = build(cfg, sd)
    g = torch.Generator().manual_seed(11)
    mix = torch.randn(3, 1500, generator=g) * 2.5 + 0.3
    std, mean = mix.std(-1, keepdim=True), mix.mean(-1, keepdim=True)
    norm = ((mix - mean) / (std + 1e-9)).unsqueeze(1)
    est = torch_oracle.forward(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, norm)
    want = est * std.unsqueeze(1) + mean.unsqueeze(1)
    if variant == "groupcomm":
        want = want + (norm - want.sum(1, keepdim=True)) / want.shape[1]
    got = pipeline.separate(model, mix.to(DEV))
    err = (got.cpu() - want).abs().max().item()
    assert err <= 1e-4, err
    with torch.no_grad():
        nrm, stats = ops.wav_normalize(mix.to(DEV).unsqueeze(1))
        unfused = ops.wav_denormalize(model(nrm), stats, nrm if variant == "groupcomm" else None)
    assert (got - unfused).abs().max().item() <= 2e-5
    assert torch.equal(pipeline.separate(model, mix.to(DEV).unsqueeze(1)), got)          # [batch, 1, time] form
    for bad in (torch.zeros(2, 2, 100, device=DEV), torch.zeros(2, 100, device=DEV)):    # the engine entry checks the shape itself
        with pytest.raises(RuntimeError):
            model._engine().separate(model, bad, False)


@pytest.mark.gpu
def test_two_stream_split_is_bit_identical(manifest):
    """The engine may run a batch as two sub-batches on two streams (examples are independent): whatever split its
    auto-tuner picks, the output equals the single-stream forward bit for bit."""
    from oracle.schema import ModelConfig
    from oracle import weights
    cfg = ModelConfig("improved", 32, 64, 2, 4, 21, 64, 2)
    model = build(cfg, weights.make_state_dict(cfg, seed=3))
    wav = torch.from_numpy(weights.make_mixture(24, 6400, seed=4)).to(DEV)
    eng = model._engine()
    with torch.no_grad():
        eng.multi_stream = False
        ref = model(wav)
        eng.multi_stream = True
        for parts in [(24,), (12, 12), (15, 9)]:
            out = torch.empty_like(ref)
            params = [p.detach() for p in model.state_dict(keep_vars=True).values()]
            eng._forward_split(parts, wav, out, eng._param_table(params, wav.device))
            torch.cuda.synchronize()
            assert torch.equal(out, ref), parts
        from sudo_rm_rf_amd import engine as engine_mod
        for _ in range(engine_mod._TUNE_AFTER + 1):          # the tune runs once the shape has come back a few times
            auto = model(wav)
            assert torch.equal(auto, ref)
        assert eng._split_choice


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cfg3_groupcomm_u8", "cfg1_improved_u8"])
def test_split_forward_stress(manifest, case):
    """Regression test for the round-1 GroupComm corruption: 200 back-to-back forwards with the batch split over two
    streams (TAC / pyramid kernels of one sub-batch co-resident with the other's MFMA GEMMs) all equal the
    single-stream forward.  Before the fix (packed bias adds in srf_tac_lanes_kernel, op_sel = 1 on src1: gfx950
    erratum) 28 of 30 such GroupComm forwards had examples off by ~5e-4.  The bar is 2e-6 max-abs, not bit
    equality: a sub-batch may be cut into different GEMM tiles than the whole batch (quarter tiles of the leftover
    round), which changes the summation order of the GlobLN statistics in the last fp64 bits."""
    from sudo_rm_rf_amd import engine as engine_mod
    cfg, sd, _, _ = load_case(manifest, case)
    model = build(cfg, sd)
    A = cfg.in_audio_channels if cfg.variant == "groupcomm" else 1
    wav = torch.from_numpy(weights_mix(32, 32000, seed=77, channels=A)).to(DEV)
    eng = model._engine()
    old = engine_mod._SPLIT_MODE
    try:
        with torch.no_grad():
            eng.multi_stream = False
            ref = model(wav).clone()
            eng.multi_stream = True
            for mode, n in (("5:3", 120), ("1:1", 80)):
                engine_mod._SPLIT_MODE = mode
                eng._split_choice.clear()
                bad, worst = 0, 0.0
                for _ in range(n):
                    err = float((model(wav) - ref).abs().max())
                    worst = max(worst, err)
                    bad += int(err > 2e-6)
                print("split %s: worst max-abs difference from the single-stream forward %.2e" % (mode, worst))
                assert bad == 0, (mode, bad, n, worst)
    finally:
        engine_mod._SPLIT_MODE = old


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cfg1_improved_u8", "cfg3_groupcomm_u8"])
def test_forward_is_reentrant_across_threads_and_streams(manifest, case):
    """The reference's forward is re-entrant (DataParallel calls it from one thread per replica; any
    caller may use its own stream).  Four threads call ONE module concurrently -- two on the default stream, two on
    their own streams -- and every result equals the sequential one bit for bit."""
    import threading
    cfg, sd, wav, _ = load_case(manifest, case)
    model = build(cfg, sd)
    # batches 8 / 9 take the two-stream split path (several C calls per forward), 2 / 3 the single srf_forward call
    xs = [torch.from_numpy(weights_mix((8, 3, 9, 2)[i], 8000 + 160 * i, seed=50 + i)).to(DEV) for i in range(4)]
    with torch.no_grad():
        want = [model(x).clone() for x in xs]
    torch.cuda.synchronize()
    got, errs = [None] * 4, []

    def work(i):
        try:
            st = torch.cuda.Stream(DEV) if i >= 2 else torch.cuda.current_stream(DEV)
            with torch.no_grad(), torch.cuda.stream(st):
                for _ in range(6):
                    y = model(xs[i])
                got[i] = y.clone()
                st.synchronize()
        except Exception as e:          # surfaced in the main thread
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(4):
        assert torch.equal(got[i], want[i]), i


def weights_mix(batch, T, seed, channels=1):
    from oracle import weights
    return weights.make_mixture(batch, T, seed=seed, channels=channels)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["improved", "groupcomm"])
def test_degenerate_inputs_match_oracle(variant):
    """Silence (zero variance in the first GlobLN: rstd = 1e4 from the 1e-8 epsilon), a constant offset and a
    large-amplitude mixture: same outputs as the oracle, relative to the output scale."""
    from oracle import weights
    from oracle.schema import ModelConfig
    cfg = (ModelConfig("improved", 16, 32, 2, 3, 21, 24, 2) if variant == "improved"
           else ModelConfig("groupcomm", 32, 64, 2, 3, 21, 24, 2, 1, 4))
    sd = weights.make_state_dict(cfg, seed=8)
    model = build(cfg, sd)
    sdt = torch_oracle.to_torch(sd)
    g = torch.Generator().manual_seed(5)
    cases = {"silence": torch.zeros(2, 1, 1600), "constant": torch.full((2, 1, 1600), 0.37),
             "loud": torch.randn(2, 1, 1600, generator=g) * 1e3, "tiny": torch.randn(2, 1, 1600, generator=g) * 1e-6}
    for name, wav in cases.items():
        with torch.no_grad():
            want = torch_oracle.forward(cfg, {k: v.double() for k, v in sdt.items()}, wav.double())
            got = model(wav.to(DEV)).cpu().double()
        assert torch.isfinite(got).all(), name
        scale = max(want.abs().max().item(), 1e-6)
        assert (got - want).abs().max().item() <= 2e-4 * scale + 1e-6, (name, (got - want).abs().max().item(), scale)


@pytest.mark.gpu
def test_small_forward_graph_replay(manifest):
    """With SRF_GRAPH=auto the engine replays a captured HIP graph after a few calls of one small (batch, T).  The
    replayed outputs equal the eager ones bit for bit, follow in-place weight updates (the graph reads the same tensors),
    and a new weight tensor (new pointer) gets a new capture."""
    from sudo_rm_rf_amd import engine as engine_mod
    cfg, sd, wav, gold = load_case(manifest, "cfg1_improved_u8")
    model = build(cfg, sd)
    x = torch.from_numpy(wav).to(DEV)
    eng = model._engine()
    old = engine_mod._GRAPH_MODE
    try:
        with torch.no_grad():
            engine_mod._GRAPH_MODE = "off"
            eager = model(x).clone()
            engine_mod._GRAPH_MODE = "auto"
            outs = [model(x).clone() for _ in range(engine_mod._GRAPH_AFTER + 3)]
            assert len(eng._graphs) == 1
            for o in outs:
                assert torch.equal(o, eager)
            x2 = torch.from_numpy(weights_mix(1, wav.shape[-1], seed=123)).to(DEV)
            engine_mod._GRAPH_MODE = "off"
            want2 = model(x2).clone()
            engine_mod._GRAPH_MODE = "auto"
            assert torch.equal(model(x2), want2)                  # same graph, new input
            model.bottleneck.bias.mul_(1.5)                       # in-place update: same pointer, graph sees the new values
            engine_mod._GRAPH_MODE = "off"
            want3 = model(x).clone()
            engine_mod._GRAPH_MODE = "auto"
            assert torch.equal(model(x), want3)
            assert np.abs(eager.cpu().numpy() - gold["out"]).max() <= TOL
    finally:
        engine_mod._GRAPH_MODE = old
