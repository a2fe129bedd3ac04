"""Host-side mirror of the reference interface: schema, attributes, pickles, error behaviour (CPU)."""
import importlib.util
import io
import os
import pickle
import re
import warnings

import numpy as np
import pytest
import torch

from oracle.schema import CONFIGS, ModelConfig, state_dict_schema

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference only exists in the build container")


def _ours(cfg):
    import sudo_rm_rf_amd.dnn.models.improved_sudormrf as imp
    import sudo_rm_rf_amd.dnn.models.groupcomm_sudormrf_v2 as gc
    return (imp.SuDORMRF if cfg.variant == "improved" else gc.GroupCommSudoRmRf)(**cfg.ctor_kwargs())


def _ref_module(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spec.loader.exec_module(mod)
    return mod


SMALL = [ModelConfig("improved", 16, 32, 2, 3, 21, 24, 2),
         ModelConfig("improved", 128, 512, 2, 4, 21, 512, 2),
         ModelConfig("groupcomm", 32, 64, 2, 3, 21, 24, 2, 1, 4),
         ModelConfig("groupcomm", 32, 64, 1, 2, 11, 16, 2, 2, 8)]


@pytest.mark.parametrize("cfg", SMALL + [CONFIGS["cfg2_improved_u16"], CONFIGS["cfg3_groupcomm_u8"]],
                         ids=lambda c: f"{c.variant}-U{c.num_blocks}-N{c.enc_num_basis}")
def test_state_dict_schema(cfg):
    m = _ours(cfg)
    sd = m.state_dict()
    schema = state_dict_schema(cfg)
    assert list(sd.keys()) == [k for k, _ in schema]
    for k, shp in schema:
        assert tuple(sd[k].shape) == shp, k
    # parameters() order == state_dict() order (the engine relies on it) and there are no buffers
    assert [id(p) for p in m.parameters()] == [id(v) for v in m.state_dict(keep_vars=True).values()]
    assert len(list(m.buffers())) == 0


def test_public_attributes_and_defaults():
    import sudo_rm_rf_amd.dnn.models.improved_sudormrf as imp
    import sudo_rm_rf_amd.dnn.models.groupcomm_sudormrf_v2 as gc
    m = imp.SuDORMRF()
    assert (m.out_channels, m.in_channels, m.num_blocks, m.upsampling_depth, m.enc_kernel_size,
            m.enc_num_basis, m.num_sources) == (128, 512, 16, 4, 21, 512, 2)
    assert m.n_least_samples_req == 10 * 16
    for name in ("encoder", "ln", "bottleneck", "sm", "mask_net", "decoder", "mask_nl_class"):
        assert hasattr(m, name)
    g = gc.GroupCommSudoRmRf(num_blocks=1)
    assert (g.in_audio_channels, g.out_channels, g.in_channels, g.upsampling_depth, g.enc_kernel_size,
            g.enc_num_basis, g.num_sources) == (1, 256, 512, 5, 21, 512, 2)
    assert g.sm[0].num_group == 16
    with pytest.raises(AssertionError):
        gc.GroupCommSudoRmRf(num_blocks=1, enc_kernel_size=20)   # odd-K assert, groupcomm_sudormrf_v2.py:255-258


def test_reference_import_paths_resolve():
    import sudo_rm_rf.dnn.experiments.utils.mixture_consistency as mixture_consistency
    import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf
    import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as sudormrf_gc_v2
    assert callable(mixture_consistency.apply)
    for name in ("SuDORMRF", "UConvBlock", "ConvNormAct", "DilatedConvNorm", "NormAct", "GlobLN"):
        assert hasattr(improved_sudormrf, name)
    for name in ("GroupCommSudoRmRf", "GC_UConvBlock", "TAC", "UConvBlock", "GlobLN"):
        assert hasattr(sudormrf_gc_v2, name)


def test_cpu_input_fails_loudly():
    from sudo_rm_rf_amd._lib import SrfError
    m = _ours(SMALL[0])
    with pytest.raises(SrfError, match="no CPU"):
        m(torch.zeros(1, 1, 400))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 400))          # 2-D input is an error in the reference as well
    import sudo_rm_rf_amd.dnn.experiments.utils.mixture_consistency as mc
    with pytest.raises(ValueError, match="Invalid mixture consistency weight type"):
        mc.apply(torch.zeros(1, 2, 8), torch.zeros(1, 1, 8), "bogus")
    with pytest.raises(RuntimeError):
        mc.apply(torch.zeros(1, 2, 8), torch.zeros(1, 1, 8))


def test_pickle_roundtrip_own_class():
    m = _ours(SMALL[2])
    m._engine()                                  # engine must not be pickled
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert "_srf_engine" not in m2.__dict__
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


@needs_ref
@pytest.mark.parametrize("cfg", SMALL, ids=lambda c: f"{c.variant}-U{c.num_blocks}")
def test_same_seed_same_weights_as_reference(cfg):
    rel = "sudo_rm_rf/dnn/models/" + ("improved_sudormrf.py" if cfg.variant == "improved" else "groupcomm_sudormrf_v2.py")
    ref = _ref_module(rel, "_ref_" + cfg.variant)
    cls = ref.SuDORMRF if cfg.variant == "improved" else ref.GroupCommSudoRmRf
    torch.manual_seed(1234)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = cls(**cfg.ctor_kwargs())
    torch.manual_seed(1234)
    o = _ours(cfg)
    rsd, osd = r.state_dict(), o.state_dict()
    assert list(rsd.keys()) == list(osd.keys())
    for k in rsd:
        assert torch.equal(rsd[k], osd[k]), k
    # load_state_dict both directions
    o.load_state_dict(rsd)
    r.load_state_dict(osd)
    # public attributes the README recipe reads back (README.md:81-88)
    for a in ("out_channels", "in_channels", "num_blocks", "upsampling_depth", "enc_kernel_size",
              "enc_num_basis", "num_sources", "n_least_samples_req"):
        assert getattr(r, a) == getattr(o, a)


@needs_ref
def test_reference_whole_module_pickle_unpickles_into_our_classes(tmp_path):
    """Published checkpoints are whole-module pickles (README.md:75): a pickle made by the REFERENCE
    classes must resolve to OUR classes when this repo is first on sys.path."""
    import subprocess
    import sys
    cfg = SMALL[2]
    path = tmp_path / "ref_model.pt"
    code = f"""
import sys, warnings, torch
sys.path.insert(0, {REF!r})
warnings.simplefilter('ignore')
import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as g
torch.manual_seed(7)
m = g.GroupCommSudoRmRf(**{cfg.ctor_kwargs()!r})
assert type(m).__module__ == 'sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2'
torch.save(m, {str(path)!r})
"""
    env = dict(os.environ, PYTHONPATH="")
    subprocess.run([sys.executable, "-c", code], check=True, cwd="/tmp", env=env)
    import sudo_rm_rf_amd.dnn.models.groupcomm_sudormrf_v2 as ours
    m = torch.load(str(path), weights_only=False)
    assert type(m) is ours.GroupCommSudoRmRf
    assert type(m.sm[0].TAC) is ours.TAC
    assert m.num_sources == cfg.num_sources and m._config_tuple()[-1] == cfg.group_size
    assert [k for k, _ in state_dict_schema(cfg)] == list(m.state_dict().keys())


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads on CPU and exports exactly what include/sudormrf_hip.h declares."""
    from sudo_rm_rf_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "sudormrf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(srf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    if not os.path.exists(_lib.LIB_PATH):
        from sudo_rm_rf_amd.build import build
        build(verbose=False)
    lib = _lib.load()
    assert lib.srf_abi_version() == _lib.ABI_VERSION
    for name in declared:
        assert hasattr(lib, name), name


def test_plan_geometry_matches_reference_padding_rule():
    """Host-only plan arithmetic (no GPU): L and T' follow improved_sudormrf.py:244,303-310."""
    import ctypes as C
    from sudo_rm_rf_amd import _lib
    from sudo_rm_rf_amd.engine import _config_struct
    lib = _lib.load()
    for cfg in SMALL + [CONFIGS["cfg2_improved_u16"], CONFIGS["cfg5_improved_u36_n4096"]]:
        for T in (1, 50, 517, 1001, 32000, 32079, 128000):
            tup = (cfg.variant, cfg.in_audio_channels, cfg.out_channels, cfg.in_channels, cfg.num_blocks,
                   cfg.upsampling_depth, cfg.enc_kernel_size, cfg.enc_num_basis, cfg.num_sources,
                   cfg.group_size)
            st = _config_struct(*tup)
            h = C.c_void_p()
            _lib.check(lib.srf_plan_create(C.byref(st), 2, T, C.byref(h)))
            assert lib.srf_plan_padded_length(h) == cfg.padded_length(T)
            assert lib.srf_plan_frames(h) == cfg.frames(T)
            assert lib.srf_plan_num_params(h) == len(state_dict_schema(cfg))
            assert lib.srf_plan_workspace_bytes(h) > 0
            lib.srf_plan_destroy(h)
    # argument validation: even kernel size, bad depth
    bad = _config_struct("improved", 1, 16, 32, 1, 3, 20, 16, 2, 1)
    h = C.c_void_p()
    assert lib.srf_plan_create(C.byref(bad), 1, 100, C.byref(h)) == -1
    assert b"odd" in lib.srf_last_error()


# ---------------------------------------------------------------------------------------------
# checkpoint tooling: DataParallel prefixes, whole-module pickles, wrapped dicts
# ---------------------------------------------------------------------------------------------
def _tiny_models():
    from sudo_rm_rf_amd.dnn.models import groupcomm_sudormrf_v2, improved_sudormrf
    torch.manual_seed(3)
    a = improved_sudormrf.SuDORMRF(out_channels=16, in_channels=32, num_blocks=2, upsampling_depth=3,
                                   enc_kernel_size=11, enc_num_basis=24, num_sources=2)
    b = groupcomm_sudormrf_v2.GroupCommSudoRmRf(in_audio_channels=1, out_channels=16, in_channels=32,
                                                num_blocks=2, upsampling_depth=3, enc_kernel_size=11,
                                                enc_num_basis=24, num_sources=2, group_size=4)
    return a, b


def test_checkpoint_module_prefix_roundtrip():
    from sudo_rm_rf_amd import checkpoint
    for m in _tiny_models():
        sd = m.state_dict()
        pref = checkpoint.add_module_prefix(sd)
        assert all(k.startswith("module.") for k in pref) and len(pref) == len(sd)
        assert list(checkpoint.strip_module_prefix(pref).keys()) == list(sd.keys())
        assert list(checkpoint.add_module_prefix(pref).keys()) == list(pref.keys())     # idempotent
        fresh = type(m)(**checkpoint.config_from_module(m))
        checkpoint.load_checkpoint({"state_dict": pref, "epoch": 3}, fresh)
        for (k, v), (k2, v2) in zip(sd.items(), fresh.state_dict().items()):
            assert k == k2 and torch.equal(v, v2)


def test_checkpoint_whole_module_pickle_and_dataparallel(tmp_path):
    from sudo_rm_rf_amd import checkpoint
    for i, m in enumerate(_tiny_models()):
        # the published pre-trained files are whole-module pickles (README.md:75-98)
        p = tmp_path / ("whole%d.pt" % i)
        torch.save(m, p)
        got = checkpoint.load_checkpoint(str(p))
        assert type(got) is type(m)
        assert checkpoint.config_from_module(got) == checkpoint.config_from_module(m)
        for v, v2 in zip(m.state_dict().values(), got.state_dict().values()):
            assert torch.equal(v, v2)
        # older runners saved the DataParallel wrapper's state_dict
        p2 = tmp_path / ("dp%d.pt" % i)
        torch.save(torch.nn.DataParallel(m).state_dict(), p2)
        fresh = torch.nn.DataParallel(type(m)(**checkpoint.config_from_module(m)))
        checkpoint.load_checkpoint(str(p2), fresh)
        for v, v2 in zip(m.state_dict().values(), fresh.module.state_dict().values()):
            assert torch.equal(v, v2)
    with pytest.raises(TypeError):
        checkpoint.load_checkpoint(m.state_dict())      # bare state_dict without a model to load into


def test_weights_of_a_data_parallel_style_replica_follow_state_dict_order():
    """torch.nn.DataParallel replicas report an empty state_dict and no parameters; the engine collects their weights
    from `_former_parameters` of every sub-module in state_dict (pre-)order.  Emulated on the CPU exactly the way
    torch/nn/parallel/replicate.py builds a replica (the GPU test runs real replicas)."""
    from collections import OrderedDict
    import torch
    import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as gc
    import sudo_rm_rf.dnn.models.improved_sudormrf as imp
    from sudo_rm_rf_amd.engine import _weights
    for model in (imp.SuDORMRF(out_channels=16, in_channels=32, num_blocks=2, upsampling_depth=3, enc_kernel_size=21,
                               enc_num_basis=24, num_sources=2),
                  gc.GroupCommSudoRmRf(in_audio_channels=1, out_channels=32, in_channels=64, num_blocks=2,
                                       upsampling_depth=3, enc_kernel_size=21, enc_num_basis=24, num_sources=2,
                                       group_size=4)):
        modules = list(model.modules())
        index = {m: i for i, m in enumerate(modules)}
        replicas = []
        for m in modules:
            r = m._replicate_for_data_parallel()
            r._former_parameters = OrderedDict()
            replicas.append(r)
        for i, m in enumerate(modules):
            for key, child in m._modules.items():
                setattr(replicas[i], key, None if child is None else replicas[index[child]])
            for key, p in m._parameters.items():
                if p is None:
                    replicas[i]._parameters[key] = None
                else:
                    copy = p.detach().clone().requires_grad_()       # stands in for the Broadcast output
                    setattr(replicas[i], key, copy)
                    replicas[i]._former_parameters[key] = copy
        rep = replicas[0]
        assert len(rep.state_dict()) == 0 and not list(rep.parameters())
        got, want = _weights(rep), list(model.state_dict(keep_vars=True).values())
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a.shape == b.shape and torch.equal(a.detach(), b.detach())
        assert [id(t) for t in _weights(model)] == [id(t) for t in want]
