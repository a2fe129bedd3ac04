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
    assert big.count("pw_conv") == 1 + 2 * 16                      # bottleneck + proj / res_conv per block
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
