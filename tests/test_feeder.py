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


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_dataset_getitem_matches_reference_golden(tmp_path, case):
    """Dataset[i] through the reference's import path == the reference's Dataset[i] (CPU float32 tensors of the same shapes):
    a batch of one through the native reader and the device normalisation kernel."""
    import sudo_rm_rf.dnn.dataset_loader.wham as wham
    c, kw = _tree(tmp_path, case)
    z = np.load(os.path.join(GOLD, case + ".npz"))
    ds = wham.Dataset(**kw)
    assert len(ds) == c["n_items"]
    for i in range(len(ds)):
        mix, src = ds[i]
        assert mix.dtype == torch.float32 and src.dtype == torch.float32 and not mix.is_cuda and not src.is_cuda
        _close(mix.numpy(), z["mix:" + ds.file_names[i]], 5e-5)
        _close(src.numpy(), z["src:" + ds.file_names[i]], 5e-5)


@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_dataset_getitem_on_a_host_without_gpu_matches_reference_golden(tmp_path, case):
    """The reference's Dataset is pure host code (wham.py:171-226); ours serves Dataset[i] on a CPU-only host too (round 6):
    native reader + the recipe on the host, against what the reference's own Dataset returned.  (On a GPU box Dataset[i]
    takes the device path -- test_dataset_getitem_matches_reference_golden -- and this test calls the host form directly.)"""
    import sudo_rm_rf.dnn.dataset_loader.wham as wham
    c, kw = _tree(tmp_path, case)
    z = np.load(os.path.join(GOLD, case + ".npz"))
    ds = wham.Dataset(**kw)
    assert len(ds) == c["n_items"]
    for i in range(len(ds)):
        mix, src = ds._getitem_host(i) if torch.cuda.is_available() else ds[i]
        assert mix.dtype == torch.float32 and src.dtype == torch.float32 and not mix.is_cuda and not src.is_cuda
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
                assert ln.shape[1] == 3 and int(ln[b].min()) == int(ln[b].max())   # (this tree: sources as long as the mixture)
                n = int(ln[b, 0])
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
        assert np.array_equal(raw[b, 0, :int(ln[b, 0])].numpy(), w[:T])
        assert abs(float(st[b, 0]) - w.astype(np.float64).mean()) <= 1e-6 * max(1.0, np.abs(w).max())


def test_batch_feeder_shards_are_disjoint_and_cover_the_set(tmp_path):
    """Rank-aware feeder (VERDICT r2 missing 2; the reference feeds all replicas from ONE DataLoader, wham.py:219-226): for
    world sizes 1 / 2 / 3 every rank builds the same epoch order; rank r's batch i is rows [r B, (r+1) B) of the
    single-process batch i of size B * world (same examples, same crops), the ranks' items of an epoch are disjoint and
    together cover what the single-process epoch covers; every rank sees the same number of batches."""
    from sudo_rm_rf_amd import _lib, feeder
    _, kw = _tree(tmp_path, "feeder_sep_clean_norm_pad")
    ds = feeder.Dataset(**dict(kw, augment=True, timelength=0.2))          # T = 1600; 7 utterances
    B = 1
    for world in (1, 2, 3):
        whole = feeder.BatchFeeder(ds, B * world, True, 3, None, 2, 11, True, host_only=True, rank=0, world_size=1)
        parts = [feeder.BatchFeeder(ds, B, True, 2, None, 2, 11, True, host_only=True, rank=r, world_size=world)
                 for r in range(world)]
        assert {len(p) for p in parts} == {len(whole)} and len(whole) == 7 // (B * world)
        for epoch in range(2):
            ref = list(whole)
            got = [list(p) for p in parts]
            items = [p.epoch_items() for p in parts]
            flat = [i for it in items for i in it]
            assert len(flat) == len(set(flat)) == len(whole) * B * world        # disjoint
            assert sorted(flat) == sorted(whole.epoch_items())                  # ... and the same set as one process
            for i, (raw, ln, st) in enumerate(ref):
                cat = torch.cat([got[r][i][0] for r in range(world)], 0)
                assert torch.equal(cat, raw)                                    # same examples, same crop starts, rank order
                assert torch.equal(torch.cat([got[r][i][1] for r in range(world)], 0), ln)
                st_cat = torch.cat([got[r][i][2] for r in range(world)], 0)      # (the 1-sample utterance has a NaN std)
                assert torch.equal(torch.nan_to_num(st_cat, nan=-7.0), torch.nan_to_num(st, nan=-7.0))
    with pytest.raises(_lib.SrfError):                                          # a sharded epoch must not end ragged
        feeder.BatchFeeder(ds, 2, True, 2, None, 2, 0, False, host_only=True, rank=0, world_size=2)
    with pytest.raises(_lib.SrfError):
        feeder.BatchFeeder(ds, 2, True, 2, None, 2, 0, True, host_only=True, rank=2, world_size=2)


def test_feeder_per_stream_lengths_and_normalize_flag(tmp_path):
    """A source file SHORTER than its mixture keeps its own valid length (the reference normalises the slice it read,
    wham.py:201-207: the zero padding must not enter the source's mean / std -- ADVICE r2); a Dataset that does not
    normalise asks for no mixture statistics (no second read of long files)."""
    from sudo_rm_rf_amd import feeder
    _, kw = _tree(tmp_path, "feeder_sep_clean_norm_pad")
    base = os.path.join(str(tmp_path), "wav8k", "min", "tr")
    rng = np.random.default_rng(5)
    short = rng.standard_normal(900).astype(np.float32)
    feeder_oracle.write_wav(os.path.join(base, "s2", "utt_00.wav"), short, 8000)      # utt_00: mixture 4000, s2 now 900
    ds = feeder.Dataset(**dict(kw, augment=False, timelength=0.25))                    # T = 2000
    bf = feeder.BatchFeeder(ds, 2, False, 2, None, 2, 0, True, host_only=True)
    raw, ln, st = next(iter(bf))
    assert ln[0].tolist() == [2000, 2000, 900] and not raw[0, 2, 900:].any()
    assert np.array_equal(raw[0, 2, :900].numpy(), short)
    ds0 = feeder.Dataset(**dict(kw, augment=False, timelength=0.25, normalize_audio=False))
    raw0, ln0, st0 = next(iter(feeder.BatchFeeder(ds0, 2, False, 2, None, 2, 0, True, host_only=True)))
    assert torch.equal(raw0, raw) and torch.equal(ln0, ln)
    assert st0.tolist() == [[0.0, 1.0], [0.0, 1.0]]                                    # untouched defaults: nothing was computed


@pytest.mark.gpu
def test_short_source_is_normalised_over_its_own_samples(tmp_path):
    """Device side of the per-stream lengths: the batch path == the oracle's recipe on the files (which slices, normalises,
    THEN pads every source on its own length)."""
    from scipy.io import wavfile
    from sudo_rm_rf_amd import feeder
    _, kw = _tree(tmp_path, "feeder_sep_clean_norm_pad")
    base = os.path.join(str(tmp_path), "wav8k", "min", "tr")
    rng = np.random.default_rng(6)
    feeder_oracle.write_wav(os.path.join(base, "s2", "utt_00.wav"), rng.standard_normal(900).astype(np.float32) + 0.3, 8000)
    ds = feeder.Dataset(**dict(kw, augment=False, timelength=0.25))
    mix, src = next(iter(ds.get_generator(batch_size=2, shuffle=False, num_workers=2)))
    waves = [np.asarray(wavfile.read(p)[1], dtype=np.float32) for p in ds.paths_of(0)]
    wm, ws = feeder_oracle.example(waves, 2000, True, True, False)
    _close(mix[0].cpu().numpy(), wm, 5e-5)
    _close(src[0].cpu().numpy(), ws, 5e-5)


@pytest.mark.gpu
def test_batch_feeder_is_stream_safe_under_an_asynchronous_consumer(tmp_path):
    """the consumer never synchronises with the host and keeps a long kernel queued per batch before it reads
    the batch; dropped batches' memory must not be re-used by the feeder's side stream while those reads are pending.  Every
    batch, copied out on the consumer stream AFTER the long kernel, must equal what the host-only feeder delivers."""
    from sudo_rm_rf_amd import feeder
    _, kw = _tree(tmp_path, "feeder_sep_clean_norm_pad")
    ds = feeder.Dataset(**dict(kw, augment=True, timelength=0.25, normalize_audio=False))   # raw copy: bitwise comparable
    dev = torch.device("cuda:0")
    want = []
    for epoch in range(6):
        host = feeder.BatchFeeder(ds, 2, True, 3, None, 2, 21, True, host_only=True)
        host._epoch = epoch
        want += [raw.clone() for raw, _, _ in host]
    gen = ds.get_generator(batch_size=2, shuffle=True, num_workers=3, device=dev, prefetch=2, seed=21)
    big = torch.randn(4096, 4096, device=dev)
    kept = []
    for epoch in range(6):
        for mix, src in gen:
            for _ in range(3):
                big = (big @ big).mul_(1e-3)                 # ~1 ms of queued consumer-stream work per batch
            kept.append(torch.cat([mix.unsqueeze(1), src], 1).clone())   # read AFTER the long kernels, on the consumer stream
            del mix, src                                     # the allocator may hand these blocks out again at once
    torch.cuda.synchronize()
    assert len(kept) == len(want) == 6 * 3
    for got, ref in zip(kept, want):
        assert torch.equal(got.cpu(), ref)


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
