"""Test infrastructure only (see oracle/__init__.py): CPU restatement of the reference's WHAM example recipe,
dataset_loader/wham.py:171-217, on numpy arrays, and a deterministic miniature WHAM tree to run it on.

Pinned to the reference itself: tools/make_golden_feeder.py builds the same tree, runs the unmodified
sudo_rm_rf.dnn.dataset_loader.wham.Dataset on it and stores what its __getitem__ returned (tests/golden/feeder_*.npz)."""
import os

import numpy as np

EPS = 1e-8
SOURCES = {"sep_clean": ("mix_clean", ["s1", "s2"]), "sep_noisy": ("mix_both", ["s1", "s2", "noise"]),
           "enh_single": ("mix_single", ["s1", "noise"])}


def write_wav(path, data, rate):
    """Minimal RIFF/WAVE writer (mono): float32 -> IEEE float, int16 -> PCM 16 (what scipy.io.wavfile.write emits)."""
    import struct
    data = np.ascontiguousarray(data)
    if data.dtype == np.float32:
        tag, bits = 3, 32
    elif data.dtype == np.int16:
        tag, bits = 1, 16
    else:
        raise TypeError(data.dtype)
    raw = data.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, tag, 1, rate, rate * bits // 8, bits // 8, bits) + b"data" + struct.pack("<I", len(raw))
    with open(path, "wb") as f:
        f.write(hdr + raw)


def make_fake_wham(root, task="sep_clean", rate=8000, split="tr", min_or_max="min", seed=0,
                   lengths=(4000, 2400, 5001, 3200, 1, 3999, 6400), int16_every=3):
    """root/wav8k/min/tr/{mixture folder, source folders}/utt_XX.wav with deterministic content; every `int16_every`-th
    utterance is stored as PCM 16 (integer magnitudes), the others as IEEE float32.  Returns the dataset directory."""
    mix_dir, srcs = SOURCES[task]
    base = os.path.join(root, "wav%dk" % (rate // 1000), min_or_max, split)
    for d in [mix_dir] + srcs:
        os.makedirs(os.path.join(base, d), exist_ok=True)
    rng = np.random.default_rng(seed)
    for i, n in enumerate(lengths):
        name = "utt_%02d.wav" % i
        parts = [(rng.standard_normal(n) * (0.05 + 0.1 * j) + 0.01 * (j + 1)).astype(np.float32) for j in range(len(srcs))]
        mix = np.sum(parts, axis=0).astype(np.float32)
        as_int = int16_every > 0 and i % int16_every == int16_every - 1
        for d, sig in zip([mix_dir] + srcs, [mix] + parts):
            if as_int:
                sig = np.clip(np.round(sig * 8000.0), -32768, 32767).astype(np.int16)
            write_wav(os.path.join(base, d, name), sig, rate)
    return base


def _normalize(x, std=None):
    """wham.py:51-55 on a 1-D float32 array (torch semantics: unbiased std, fp32 arithmetic)."""
    x = x.astype(np.float32)
    mean = np.float32(x.mean(dtype=np.float64))
    if std is None:
        std = np.float32(x.std(ddof=1, dtype=np.float64)) if x.size > 1 else np.float32(np.nan)
    return ((x - mean) / (np.float32(std) + np.float32(EPS))).astype(np.float32)


def _safe_pad(x, T, zero_pad):
    """wham.py:157-166."""
    if zero_pad and x.shape[0] < T:
        out = np.zeros(T, dtype=np.float32)
        out[:x.shape[0]] = x
        return out
    return x[:T]


def example(waves, T, normalize_audio, zero_pad, augment, rand_start=0):
    """waves: [mixture, source 1, ...] full-length float32 arrays as read from the files -> (mixture [T], sources [S, T]),
    wham.py:171-217 with the random crop start given."""
    mix = waves[0]
    if augment and mix.shape[0] > T:
        mix = mix[rand_start:rand_start + T]                                  # :183-186
    else:
        rand_start = 0
    if normalize_audio:
        mix = _normalize(mix)                                                 # :190-191
    mix = _safe_pad(mix, T, zero_pad)                                         # :192
    srcs = []
    for w in waves[1:]:
        s = w[rand_start:rand_start + T]                                      # :201
        if normalize_audio:
            s = _normalize(s)                                                 # :205-207
        srcs.append(_safe_pad(s, T, zero_pad))
    if normalize_audio:
        mix_std = np.float32(mix.std(dtype=np.float64))                       # :212 (numpy: population std)
        mix = _normalize(mix, std=mix_std)                                    # :213
        srcs = [_normalize(s, std=mix_std) for s in srcs]                     # :214-215
    return mix.astype(np.float32), np.stack(srcs).astype(np.float32)
