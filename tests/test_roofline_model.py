"""CPU checks of sudo_rm_rf_amd/roofline.py: the byte / flop models bench.py reports against and the
launch model it pairs with the in-library profiler's marks (one entry per kernel launch of srf_forward, in launch order)."""
import pytest

from sudo_rm_rf_amd import roofline

CFG2 = dict(variant="improved", B=256, C=512, U=16, D=5, K=21, N=512, S=2, T=32000)
CFG3 = dict(variant="groupcomm", B=256, C=512, U=8, D=5, K=21, N=512, S=2, T=32000, G=16)


def names(**kw):
    return [n for n, _, _ in roofline.launch_model(**kw)]


def test_survey_byte_and_flop_figures():
    """cfg 2 = 1206.2 MB and 30.2 GFLOP per example."""
    kw = dict(CFG2)
    assert roofline.bytes_per_example(**kw) / 1e6 == pytest.approx(1206.2, abs=0.2)
    assert roofline.flops_per_example(**kw) / 1e9 == pytest.approx(30.2, abs=0.1)


def test_launch_model_follows_the_dispatch():
    big = names(Bt=32, **CFG2)
    # fused tail at bench batch: pack, one GEMM launch, overlap-add -- no masked tensor, no frame GEMM
    assert big[-3:] == ["pack_decoder", "pw_mask_decode", "overlap_add"] and "transpose" not in big
    assert big.count("pyramid_moments") == big.count("pyramid_finalize") == big.count("pyramid_merge") == 16
    # round 5: bottleneck + proj_1x1 of block 0, res_conv of block i + proj_1x1 of block i + 1 as ONE launch each; the last res_conv alone
    assert big.count("pw_pair") == 16 and big.count("pw_conv") == 1
    sep = [n for n, _, _ in roofline.launch_model(Bt=32, pairs=False, **CFG2)]
    assert sep.count("pw_conv") == 1 + 2 * 16 and "pw_pair" not in sep      # debug flag 1: bottleneck + proj / res_conv per block
    paired_bytes = sum(b for _, b, _ in roofline.launch_model(Bt=32, **CFG2))
    sep_bytes = sum(b for _, b, _ in roofline.launch_model(Bt=32, pairs=False, **CFG2))
    assert sep_bytes - paired_bytes == 16 * 4 * 32 * 3200 * 256           # proj_1x1 no longer re-reads the 256-channel tensor
    cfg4 = dict(CFG2, B=512, U=36, D=6, N=2048)
    assert "pw_pair" not in [n for n, _, _ in roofline.launch_model(Bt=32, **cfg4)]     # B = 512: no block holds all of conv 2's k
    small = names(Bt=1, **dict(CFG2, U=8))
    # batch 1: fewer 256 x 128 tiles than CUs -> the tail stays unfused (mask GEMM, transpose, zero bias, frame GEMM)
    assert small[-5:] == ["pw_conv", "transpose", "zero_fill", "pw_conv", "overlap_add"]
    unfused = [n for n, _, _ in roofline.launch_model(Bt=32, fuse_tail=False, **CFG2)]
    assert unfused[-5:] == ["pw_conv", "transpose", "zero_fill", "pw_conv", "overlap_add"]
    gc = names(Bt=32, **CFG3)
    # GroupComm: the TAC's norm + residual is folded into the per-group proj conv (no gln_apply_add launch)
    assert gc.count("tac") == 8 and "gln_apply_add" not in gc and gc.count("pw_conv") == 1 + 2 * 8
    generic = names(Bt=32, kernel_mode=1, **CFG3)
    assert generic.count("gln_apply_add") == 8


def test_launch_model_bytes_are_positive_and_tail_fusion_removes_traffic():
    fused = sum(b for _, b, _ in roofline.launch_model(Bt=32, **CFG2))
    unfused = sum(b for _, b, _ in roofline.launch_model(Bt=32, fuse_tail=False, **CFG2))
    assert 0 < fused < unfused and unfused - fused > 0.7e9          # the masked tensor's write + read (2 x 419 MB) minus the partials


def test_training_byte_and_flop_model():
    """The training step's models: parameter counts = the README's, bytes / FLOPs on the documented rules,
    and every family of the launch model carries positive bytes."""
    from sudo_rm_rf_amd import roofline as R
    d2 = dict(variant="improved", B=256, C=512, U=16, D=5, K=21, N=512, S=2, T=32000)
    d4 = dict(variant="improved", B=512, C=512, U=36, D=6, K=21, N=2048, S=2, T=32000)
    d3 = dict(variant="groupcomm", B=256, C=512, U=8, D=5, K=21, N=512, S=2, T=32000, G=16)
    pk = lambda d: {k: v for k, v in d.items() if k != "T"}
    assert R.n_params(**pk(d2)) == 5016353 and R.n_params(**pk(d4)) == 23239241 and R.n_params(**pk(d3)) == 507177
    for d in (d2, d4, d3):
        fwd, trn = R.bytes_per_example(**d), R.train_bytes_per_example(**d)
        assert 2.0 * fwd < trn < 3.0 * fwd                    # two mirrored passes + the re-reads of the saved activations
        assert R.train_flops_per_example(**d) == 3.0 * R.flops_per_example(**d)
    L = R.frames(32000, 21, 5)
    want = 2 * R.bytes_per_example(**d2) + 4.0 * (512 * L + 16 * (256 * L + (4 - 2.0 ** -4) * 512 * L) + 256 * L) + 4.0 * 3 * 2 * 512 * L
    assert abs(R.train_bytes_per_example(**d2) - want) < 1.0
    fam = R.train_family_model(Bt=32, **d2)
    fam3 = R.train_family_model(Bt=32, **d3)              # GroupComm (round 4): its own block families, the shared head / tail
    for name in ("tac_mfma", "tac_bwd_mfma", "pw_conv_small", "pw_wgrad_small", "dwconv5_bwd", "gln_bwd_apply", "pyramid_merge_save"):
        assert fam3[name][0] > 0, name
    assert not any(k.startswith("pw_conv_x3w4<0>") or k.startswith("pw_conv_x3w4<2>") for k in fam3)     # no block GEMMs on the MFMA kernels
    # the TAC backward writes the operand tensors of its weight gradients: Z and GPZ ([Bt G, H, L]) dominate its bytes
    assert fam3["tac_bwd_mfma"][0] > 8 * 4.0 * 32 * L * 2 * 16 * 48
    ks3 = sum(b for b, _ in fam3.values())
    assert 1.0 < ks3 / (32 * R.train_bytes_per_example(**d3)) < 2.5
    for name in ("pw_pair_x3f4<1>", "pw_pair_x3f4<2>", "pw_conv_x3w4<2>", "pw_conv_x3w<0>", "pw_wgrad", "dwconv5_bwd", "gln_bwd_apply",
                 "gln_bwd_reduce", "pyramid_merge_save", "pyramid_moments", "clip_adam"):
        assert fam[name][0] > 0, name
    assert "pw_conv_x3w4<0>" not in fam and "pw_conv_x3w4<1>" not in fam      # (cfg 2: every proj_1x1 and the bottleneck ride in a pair)
    # round 5: U - 1 data-gradient pairs at the bench shape (B = 256, 800 tiles per launch); none at cfg 4 (B = 512) or at batch 4
    assert fam["pw_pair_x3f<0>"][0] == 15 * 4.0 * 32 * L * (512 + 2 * 256 + 512)
    assert "pw_pair_x3f<0>" not in R.train_family_model(Bt=32, **d4) and "pw_pair_x3f<0>" not in R.train_family_model(Bt=4, **d2)
    unpaired = R.train_family_model(Bt=32, dgrad_pairs=False, **d2)
    # what the pairs save: the re-read of g_x (15 backward pairs) and of the 256-channel residual stream (15 + 1 forward pairs)
    assert sum(b for b, _ in unpaired.values()) - sum(b for b, _ in fam.values()) == (15 + 16) * 4.0 * 32 * L * 256
    assert sum(fl for _, fl in unpaired.values()) == sum(fl for _, fl in fam.values())
    # the forward GEMMs' bytes are the inference launch model's (same tensors): U x 4 Bt L (B + C) for proj_1x1
    assert unpaired["pw_conv_x3w4<0>"][0] == 16 * 4.0 * 32 * L * (256 + 512)
    assert fam["pw_pair_x3f4<2>"][0] == 15 * 4.0 * 32 * L * (512 + 2 * 256 + 512)
    # what this kernel set must move is more than the fusion-minimal figure (saved levels, separate norm passes), but < 2 x it
    ks = sum(b for b, _ in fam.values())
    assert 1.0 < ks / (32 * R.train_bytes_per_example(**d2)) < 2.0
