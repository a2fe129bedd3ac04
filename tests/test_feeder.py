"""Input feeder: native WAV reader + batch feeder (csrc/srf_feeder.hip, sudo_rm_rf_amd/feeder.py)
against fixtures produced by the reference's own Dataset on a deterministic miniature WHAM tree
(tools/make_golden_feeder.py -> tests/golden/feeder_*.npz).  Host side on the CPU; the device normalisation under -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import feeder_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "FEEDER_MANIFEST.json")))


def _tree(tmp_path, case):
    c = MANIFEST[case]
    feeder_oracle.make_fake_wham(str(tmp_path), task=c["task"], seed=c["seed"])
    return c, dict(root_dirpath=str(tmp_path), task=c["task"], split="tr", sample_rate=8000, timelength=c["timelength"],
                   normalize_audio=c["normalize_audio"], n_samples=0, zero_pad=c["zero_pad"], augment=c["augment"],
                   min_or_max="min")


def _close(got, want, tol=2e-5):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    assert (np.isnan(got) == np.isnan(want)).all()
    m = ~np.isnan(want)
    scale = max(1.0, np.abs(want[m]).max()) if m.any() else 1.0
    assert np.abs(got[m] - want[m]).max() <= tol * scale if m.any() else True


@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_feeder_oracle_matches_reference_golden(tmp_path, case):
    """oracle/feeder_oracle.example (files read with scipy) == what the reference's Dataset.__getitem__ returned."""
    from scipy.io import wavfile
    c, kw = _tree(tmp_path, case)
    z = np.load(os.path.join(GOLD, case + ".npz"))
    mix_dir, srcs = feeder_oracle.SOURCES[c["task"]]
    base = os.path.join(str(tmp_path), "wav8k", "min", "tr")
    T = int(8000 * c["timelength"])
    names = sorted(k[4:] for k in z.files if k.startswith("mix:"))
    assert len(names) == c["n_items"]
    for name in names:
        waves = [np.asarray(wavfile.read(os.path.join(base, d, name))[1], dtype=np.float32) for d in [mix_dir] + srcs]
        mix, src = feeder_oracle.example(waves, T, c["normalize_audio"], c["zero_pad"], c["augment"])
        _close(mix, z["mix:" + name])
        _close(src, z["src:" + name])


def test_native_wav_reader(tmp_path):
    from sudo_rm_rf_amd import _lib, feeder
    rng = np.random.default_rng(0)
    f32 = rng.standard_normal(1000).astype(np.float32)
    i16 = (rng.standard_normal(777) * 3000).astype(np.int16)
    feeder_oracle.write_wav(str(tmp_path / "a.wav"), f32, 8000)
    feeder_oracle.write_wav(str(tmp_path / "b.wav"), i16, 16000)
    assert feeder.wav_info(str(tmp_path / "a.wav")) == (8000, 1, 32, 1000)
    assert feeder.wav_info(str(tmp_path / "b.wav")) == (16000, 1, 16, 777)
    assert np.array_equal(feeder.wav_read(str(tmp_path / "a.wav")), f32)
    assert np.array_equal(feeder.wav_read(str(tmp_path / "b.wav")), i16.astype(np.float32))     # integer magnitudes, unscaled
    assert np.array_equal(feeder.wav_read(str(tmp_path / "a.wav"), 990, 50), f32[990:])          # clipped at the end
    assert feeder.wav_read(str(tmp_path / "a.wav"), 2000, 5).size == 0
    from scipy.io import wavfile                                                                  # and scipy-written files
    wavfile.write(str(tmp_path / "c.wav"), 8000, i16)
    assert np.array_equal(feeder.wav_read(str(tmp_path / "c.wav")), i16.astype(np.float32))
    (tmp_path / "junk.wav").write_bytes(b"not a wav file at all")
    with pytest.raises(_lib.SrfError):
        feeder.wav_info(str(tmp_path / "junk.wav"))
    with pytest.raises(_lib.SrfError):
        feeder.wav_read(str(tmp_path / "missing.wav"))


@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_dataset_getitem_matches_reference_golden(tmp_path, case):
    """Dataset[i] through the reference's import path (native file reader) == the reference's Dataset[i]."""
    import sudo_rm_rf.dnn.dataset_loader.wham as wham
    c, kw = _tree(tmp_path, case)
    z = np.load(os.path.join(GOLD, case + ".npz"))
    ds = wham.Dataset(**kw)
    assert len(ds) == c["n_items"]
    for i in range(len(ds)):
        mix, src = ds[i]
        assert mix.dtype == torch.float32 and src.dtype == torch.float32
        _close(mix.numpy(), z["mix:" + ds.file_names[i]])
        _close(src.numpy(), z["src:" + ds.file_names[i]])


def test_dataset_argument_checks(tmp_path):
    import sudo_rm_rf.dnn.dataset_loader.wham as wham
    _, kw = _tree(tmp_path, "feeder_sep_clean_norm_pad")
    with pytest.raises(KeyError):
        wham.Dataset(**{k: v for k, v in kw.items() if k != "augment"})
    with pytest.raises(ValueError):
        wham.Dataset(**dict(kw, task="no_such_task"))
    with pytest.raises(TypeError):
        wham.Dataset(**dict(kw, timelength=4))
    with pytest.raises(IOError):
        wham.Dataset(**dict(kw, split="tt"))


def test_batch_feeder_host_side(tmp_path):
    """The native batch path without a GPU: every item once per epoch, crops shared by an example's files, lengths, padding,
    mixture statistics over the range the reference normalises on, reshuffle per epoch, drop_last."""
    from sudo_rm_rf_amd import feeder
    _, kw = _tree(tmp_path, "feeder_sep_clean_norm_pad")
    ds = feeder.Dataset(**dict(kw, augment=True, timelength=0.3))          # T = 2400; files of 1 .. 6400 samples
    T = ds.time_samples
    bf = feeder.BatchFeeder(ds, 3, True, 4, None, 2, 5, False, host_only=True)
    assert len(bf) == 3
    full = {n: [feeder.wav_read(p) for p in ds.paths_of(i)] for i, n in enumerate(ds.file_names)}
    orders = []
    for epoch in range(2):
        seen = []
        for raw, ln, st in bf:
            assert raw.shape[1:] == (3, T)
            for b in range(raw.shape[0]):
                r = raw[b].numpy()
                # which item is it, and where was it cropped?  (the mixture identifies both; the sources must follow)
                n = int(ln[b])
                found = []
                for k, wk in full.items():
                    L = len(wk[0])
                    if L <= T:
                        if n == L and np.array_equal(r[0][:n], wk[0]):
                            found.append((k, 0))
                    elif n == T:
                        first = np.flatnonzero(wk[0][:L - T + 1] == r[0][0])
                        found += [(k, int(s)) for s in first if np.array_equal(wk[0][s:s + T], r[0])]
                assert len(found) == 1
                name, s0 = found[0]
                w = full[name]
                seen.append(name)
                crop = w[0][s0:s0 + T]
                if len(w[0]) > T:
                    assert 0 <= s0 < len(w[0]) - T
                for j in range(3):
                    assert np.array_equal(r[j][:n], w[j][s0:s0 + T])
                    assert not r[j][n:].any()
                assert abs(float(st[b, 0]) - crop.astype(np.float64).mean()) <= 1e-6 * max(1.0, np.abs(crop).max())
                if n > 1:
                    assert abs(float(st[b, 1]) - crop.astype(np.float64).std(ddof=1)) <= 1e-5 * max(1.0, np.abs(crop).max())
        assert sorted(seen) == sorted(ds.file_names)
        orders.append(seen)
    assert orders[0] != orders[1]                      # reshuffled
    # without the random crop the mixture statistics cover the WHOLE file (the reference truncates after normalising)
    ds2 = feeder.Dataset(**dict(kw, augment=False, timelength=0.3))
    bf2 = feeder.BatchFeeder(ds2, 4, False, 2, None, 3, 0, True, host_only=True)
    assert len(bf2) == 1
    (raw, ln, st), = list(bf2)
    assert raw.shape[0] == 4
    for b in range(4):
        w = full[ds2.file_names[b]][0]
        assert np.array_equal(raw[b, 0, :int(ln[b])].numpy(), w[:T])
        assert abs(float(st[b, 0]) - w.astype(np.float64).mean()) <= 1e-6 * max(1.0, np.abs(w).max())


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_batch_feeder_on_device_matches_reference_golden(tmp_path, case):
    """get_generator on the MI355X (native threads -> pinned buffer -> async copy -> srf_feeder_normalize) == the reference's
    Dataset[i], example by example."""
    import sudo_rm_rf.dnn.dataset_loader.wham as wham
    c, kw = _tree(tmp_path, case)
    z = np.load(os.path.join(GOLD, case + ".npz"))
    ds = wham.Dataset(**kw)
    gen = ds.get_generator(batch_size=2, shuffle=False, num_workers=3)
    i = 0
    for mix, src in gen:
        assert mix.is_cuda and src.is_cuda and mix.shape[0] == 2
        for b in range(mix.shape[0]):
            _close(mix[b].cpu().numpy(), z["mix:" + ds.file_names[i]], 5e-5)
            _close(src[b].cpu().numpy(), z["src:" + ds.file_names[i]], 5e-5)
            i += 1
    assert i == 2 * (len(ds) // 2)                    # drop_last=True, as the reference's DataLoader
    # a second epoch, shuffled, still delivers whole batches of known examples
    gen2 = ds.get_generator(batch_size=3, shuffle=True, num_workers=2, seed=3)
    n = sum(m.shape[0] for m, _ in gen2)
    assert n == 3 * (len(ds) // 3)
