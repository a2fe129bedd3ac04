"""Input feeder: the reference's WHAM ``Dataset`` + ``get_generator`` (dataset_loader/wham.py:58-226)
with the per-example work moved off Python.

``Dataset(**kwargs)`` takes the reference's keyword arguments and indexes the same directory tree.  ``get_generator`` returns
an iterator over ``(mixtures [B, T], sources [B, S, T])`` float32 tensors ON THE MI355X: a pool of native threads
(csrc/srf_feeder.hip) reads and crops the batch's WAV files into pinned buffers, one asynchronous copy moves the raw batch,
and one kernel applies the Dataset's normalisation recipe there.  ``Dataset[i]`` (one example, CPU tensors, the reference's
``__getitem__``) is kept for API compatibility and as the parity reference of the batch path."""
import ctypes as C
import glob
import os
from time import time

import numpy as np
import torch

from . import _lib

EPS = 1e-8
# wham.py:24-48
WHAM_TASKS = {
    "enhance_single_white_noise": {"mixture": "source_with_white_noise", "sources": ["s1", "white_noise"], "n_sources": 1},
    "enhance_single": {"mixture": "mix_single", "sources": ["s1", "noise"], "n_sources": 1},
    "enhance_both": {"mixture": "mix_both", "sources": ["mix_clean", "noise"], "n_sources": 1},
    "sep_clean": {"mixture": "mix_clean", "sources": ["s1", "s2"], "n_sources": 2},
    "sep_noisy": {"mixture": "mix_both", "sources": ["s1", "s2", "noise"], "n_sources": 2},
}
WHAM_TASKS["enh_single"] = WHAM_TASKS["enhance_single"]
WHAM_TASKS["enh_both"] = WHAM_TASKS["enhance_both"]


def normalize_tensor_wav(wav_tensor, eps=1e-8, std=None):
    """wham.py:51-55."""
    mean = wav_tensor.mean(-1, keepdim=True)
    if std is None:
        std = wav_tensor.std(-1, keepdim=True)
    return (wav_tensor - mean) / (std + eps)


def wav_info(path):
    """(sample rate, channels, bits per sample, samples per channel) of a WAV file, through the native reader."""
    lib = _lib.load()
    r, ch, bits, fr = C.c_int(), C.c_int(), C.c_int(), C.c_long()
    _lib.check(lib.srf_wav_info(os.fsencode(path), C.byref(r), C.byref(ch), C.byref(bits), C.byref(fr)), "srf_wav_info")
    return r.value, ch.value, bits.value, fr.value


def wav_read(path, start=0, count=None):
    """Samples [start, start + count) of a mono WAV file as float32 (values as scipy.io.wavfile.read +
    torch.tensor(dtype=float32) deliver them), through the native reader."""
    lib = _lib.load()
    if count is None:
        count = wav_info(path)[3] - start
    out = np.empty(max(count, 0), dtype=np.float32)
    fr = C.c_long()
    _lib.check(lib.srf_wav_read(os.fsencode(path), int(start), int(out.size), out.ctypes.data_as(C.c_void_p), C.byref(fr)),
               "srf_wav_read")
    return out[:max(0, min(out.size, fr.value - start))]


class Dataset(torch.utils.data.Dataset):
    """Mirror of dataset_loader/wham.py:58-226.  kwargs (all required, as in the reference): root_dirpath, task, split,
    sample_rate, timelength, normalize_audio, n_samples, zero_pad, augment, min_or_max.

    Differences, all outside the arithmetic: the file index is sorted by name (the reference keeps glob2's order), the
    lengths come from the WAV headers (no ``metadata`` pickle is written into the dataset directory), and the random crop of
    the batch path draws from a seeded splitmix64 stream instead of a time-seeded numpy generator."""

    def __init__(self, **kwargs):
        super().__init__()
        self.kwargs = kwargs

        def arg(key, typ, choices=None, check=None):
            if key not in kwargs:
                raise KeyError("Argument: <{}> does not exist in pytorch dataloader keyword arguments".format(key))
            v = kwargs[key]
            if not isinstance(v, typ):
                raise TypeError("Value: <{}> for key: <{}> is not an instance of the known selected type: <{}>".format(v, key, typ))
            if choices is not None and v not in choices:
                raise ValueError("Value: <{}> for key: <{}> is not in the regime of the appropriate choices: <{}>".format(
                    v, key, list(choices)))
            if check is not None and not check(v):
                raise ValueError("Value(s): <{}> for key: <{}> does/do not fulfill the predefined checks".format(v, key))
            return v

        self.task = arg("task", str, WHAM_TASKS.keys())
        self.zero_pad = arg("zero_pad", bool)
        self.augment = arg("augment", bool)
        self.normalize_audio = arg("normalize_audio", bool)
        self.min_or_max = arg("min_or_max", str, ["min", "max"])
        self.split = arg("split", str, ["cv", "tr", "tt"])
        self.n_samples = arg("n_samples", int, check=lambda x: x >= 0)
        self.sample_rate = arg("sample_rate", int)
        self.root_path = arg("root_dirpath", str, check=os.path.lexists)
        self.dataset_dirpath = self.get_path()
        self.timelength = arg("timelength", float)
        self.time_samples = int(self.sample_rate * self.timelength)

        mix_folder = os.path.join(self.dataset_dirpath, WHAM_TASKS[self.task]["mixture"])
        info = []
        for path in sorted(glob.glob(os.path.join(mix_folder, "*.wav"))):
            rate, _, _, frames = wav_info(path)
            assert rate == self.sample_rate
            info.append((os.path.basename(path), frames))
        self.mixtures_info = info
        names = [(p, n) for (p, n) in info if (n >= self.time_samples or self.zero_pad)]
        if self.n_samples > 0:
            names = names[:self.n_samples]
        if not names:
            raise IOError("no usable mixtures under {}".format(mix_folder))
        max_time_samples = max(n for _, n in names)
        self.file_names = [p for p, _ in names]
        self.file_frames = [n for _, n in names]
        if self.time_samples <= 0.:          # "the whole audio input"
            self.time_samples = max_time_samples
        self.source_names = list(WHAM_TASKS[self.task]["sources"])

    def get_path(self):
        path = os.path.join(self.root_path, "wav{}k".format(int(self.sample_rate / 1000)), self.min_or_max, self.split)
        if os.path.lexists(path):
            return path
        raise IOError("Dataset path: {} not found!".format(path))

    def safe_pad(self, tensor_wav):
        """wham.py:157-166."""
        if self.zero_pad and tensor_wav.shape[0] < self.time_samples:
            padded = torch.zeros(list(tensor_wav.shape[:-1]) + [self.time_samples], dtype=torch.float32)
            padded[:tensor_wav.shape[0]] = tensor_wav
            return padded[:self.time_samples]
        return tensor_wav[:self.time_samples]

    def __len__(self):
        return len(self.file_names)

    def paths_of(self, idx):
        name = self.file_names[idx]
        return [os.path.join(self.dataset_dirpath, WHAM_TASKS[self.task]["mixture"], name)] + \
               [os.path.join(self.dataset_dirpath, s, name) for s in self.source_names]

    def __getitem__(self, idx):
        """One example on the CPU, operation for operation wham.py:171-217 (files through the native reader)."""
        if self.augment:
            np.random.seed(int(np.modf(time())[0] * 100000000))
        paths = self.paths_of(idx)
        max_len = self.file_frames[idx]
        rand_start = 0
        if self.augment and max_len > self.time_samples:
            rand_start = np.random.randint(0, max_len - self.time_samples)
            mixture_wav = torch.from_numpy(wav_read(paths[0], rand_start, self.time_samples).copy())
        else:
            mixture_wav = torch.from_numpy(wav_read(paths[0]).copy())
        if self.normalize_audio:
            mixture_wav = normalize_tensor_wav(mixture_wav)
        mixture_wav = self.safe_pad(mixture_wav)
        sources_list = []
        for p in paths[1:]:
            source_wav = torch.from_numpy(wav_read(p, rand_start, self.time_samples).copy())
            if self.normalize_audio:
                source_wav = normalize_tensor_wav(source_wav)
            sources_list.append(self.safe_pad(source_wav))
        if self.normalize_audio:
            mix_std = mixture_wav.detach().cpu().numpy().std()
            mixture_wav = normalize_tensor_wav(mixture_wav, std=mix_std)
            sources_list = [normalize_tensor_wav(s, std=mix_std) for s in sources_list]
        return mixture_wav, torch.stack(sources_list, dim=0)

    def get_generator(self, batch_size=4, shuffle=True, num_workers=4, device=None, prefetch=3, seed=0, drop_last=True):
        """The reference's DataLoader(batch_size, shuffle, num_workers, drop_last=True) (wham.py:219-224) as a native
        feeder: iterating it yields (mixtures [B, T], sources [B, S, T]) on `device` (default: the current MI355X)."""
        return BatchFeeder(self, batch_size, shuffle, num_workers, device, prefetch, seed, drop_last)


class BatchFeeder:
    """Iterable over the batches of one epoch per ``iter()`` (epoch counter advances, so every epoch reshuffles / recrops).
    Pipeline per batch: native threads fill a pinned buffer -> cudaMemcpyAsync on a side stream -> srf_feeder_normalize on
    that stream -> the consumer's stream waits on the batch's event.  `prefetch` batches are in flight."""

    def __init__(self, dataset, batch_size, shuffle, num_workers, device, prefetch, seed, drop_last, host_only=False):
        if not dataset.zero_pad and any(n < dataset.time_samples for n in dataset.file_frames):
            raise _lib.SrfError("files shorter than time_samples need zero_pad=True to be batched")
        self.ds, self.B, self.host_only = dataset, int(batch_size), host_only
        self.S1 = 1 + len(dataset.source_names)
        self.T = dataset.time_samples
        self.prefetch = max(1, int(prefetch))
        lib = _lib.load()
        flat = [os.fsencode(p) for i in range(len(dataset)) for p in dataset.paths_of(i)]
        arr = (C.c_char_p * len(flat))(*flat)
        h = C.c_void_p()
        _lib.check(lib.srf_feeder_create(arr, len(dataset), self.S1, self.T, self.B, max(1, int(num_workers)),
                                         int(dataset.augment), int(bool(shuffle)), int(bool(drop_last)), int(seed), C.byref(h)),
                   "srf_feeder_create")
        self._h, self._lib, self._epoch = h, lib, 0
        if host_only:
            self.device = None
            mk = lambda shape, dt: torch.empty(shape, dtype=dt)
        else:
            self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
            if self.device.type != "cuda":
                raise _lib.SrfError("BatchFeeder delivers to an MI355X only (got %s)" % self.device)
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
            self._stream = torch.cuda.Stream(self.device)
        self._slots = [(mk((self.B, self.S1, self.T), torch.float32), mk((self.B,), torch.int32), mk((self.B, 2), torch.float32))
                       for _ in range(self.prefetch)]

    def __len__(self):
        return int(self._lib.srf_feeder_batches_per_epoch(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.srf_feeder_destroy(h)

    def _submit(self, slot):
        w, n, st = self._slots[slot]
        rc = self._lib.srf_feeder_submit(self._h, w.data_ptr(), n.data_ptr(), st.data_ptr())
        if rc == 1:
            return False
        _lib.check(rc, "srf_feeder_submit")
        return True

    def __iter__(self):
        lib = self._lib
        _lib.check(lib.srf_feeder_start_epoch(self._h, self._epoch), "srf_feeder_start_epoch")
        self._epoch += 1
        queued = []
        for s in range(self.prefetch):
            if self._submit(s):
                queued.append(s)
        try:
            while queued:
                slot = queued.pop(0)
                nv = C.c_int()
                _lib.check(lib.srf_feeder_wait(self._h, None, None, None, C.byref(nv)), "srf_feeder_wait")
                w, n, st = self._slots[slot]
                if self.host_only:
                    out = (w[:nv.value].clone(), n[:nv.value].clone(), st[:nv.value].clone())
                else:
                    out = self._to_device(w, n, st, nv.value)
                if self._submit(slot):      # the slot's buffers are free again (the copy above is complete / ordered)
                    queued.append(slot)
                yield out
        finally:
            while queued:                   # abandoned iteration: drain what is in flight
                queued.pop(0)
                lib.srf_feeder_wait(self._h, None, None, None, None)

    def _to_device(self, w, n, st, nv):
        dev, lib = self.device, self._lib
        mix = torch.empty((self.B, self.T), dtype=torch.float32, device=dev)
        src = torch.empty((self.B, self.S1 - 1, self.T), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev), torch.cuda.stream(self._stream):
            raw = w.to(dev, non_blocking=True)
            nd = n.to(dev, non_blocking=True)
            sd = st.to(dev, non_blocking=True)
            _lib.check(lib.srf_feeder_normalize(raw.data_ptr(), nd.data_ptr(), sd.data_ptr(), self.B, self.S1, self.T,
                                                int(self.ds.normalize_audio), C.c_float(EPS), mix.data_ptr(), src.data_ptr(),
                                                _lib.current_stream(dev)), "srf_feeder_normalize")
            done = torch.cuda.Event()
            done.record(self._stream)
        done.synchronize()                  # the pinned slot is resubmitted right after: its copy must have left the host
        for t in (mix, src, raw, nd, sd):
            t.record_stream(torch.cuda.current_stream(dev))
        return mix[:nv], src[:nv]
