"""Input feeder: the reference's WHAM ``Dataset`` + ``get_generator`` (dataset_loader/wham.py:58-226)
with the per-example work moved off Python.

``Dataset(**kwargs)`` takes the reference's keyword arguments and indexes the same directory tree.  ``get_generator`` returns
an iterator over ``(mixtures [B, T], sources [B, S, T])`` float32 tensors ON THE MI355X: a pool of native threads
(csrc/srf_feeder.hip) reads and crops the batch's WAV files into pinned buffers, one asynchronous copy moves the raw batch,
and one kernel applies the Dataset's normalisation recipe there; with one process per GPU every rank's feeder deals itself a
disjoint shard of each global batch (``rank`` / ``world_size``).  ``Dataset[i]`` (the reference's ``__getitem__``: one example,
CPU tensors) is a batch of one through the same native reader and device kernel -- or, on a host without a GPU, the native
reader plus the recipe on the host (``normalize_tensor_wav`` / ``safe_pad``, the mirrors of the reference's public helpers);
nothing on the batch feeder's path calls those two."""
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
        """One example, wham.py:171-217: (mixture [T], sources [S, T]) float32 CPU tensors like the reference returns them.
        With an MI355X visible: a batch of one through the batch feeder's path -- the native reader (srf_feeder_read_example)
        and its device kernel (srf_feeder_normalize).  On a host WITHOUT a GPU (round 6; the reference's Dataset is pure host
        code and is used that way, e.g. to inspect or pre-process a corpus): the same native reader and the recipe's
        arithmetic on the host (`_getitem_host`) -- data-loader glue, not the model's hot path, which has no CPU form.
        The random crop start is drawn like the reference's (time-seeded numpy generator, :173-186)."""
        if self.augment:
            np.random.seed(int(np.modf(time())[0] * 100000000))
        paths = self.paths_of(idx)
        max_len = self.file_frames[idx]
        rand_start = 0
        if self.augment and max_len > self.time_samples:
            rand_start = np.random.randint(0, max_len - self.time_samples)
        if not torch.cuda.is_available():
            return self._getitem_host(idx, rand_start)
        # what the reference's safe_pad leaves: time_samples with zero_pad, else at most the samples the file has (:157-166)
        T = self.time_samples if self.zero_pad else max(1, min(self.time_samples, max_len - rand_start))
        S1 = len(paths)
        wave = torch.empty((1, S1, T), dtype=torch.float32).pin_memory()
        ln = torch.empty((1, S1), dtype=torch.int32)
        st = torch.empty((1, 2), dtype=torch.float32)
        arr = (C.c_char_p * S1)(*[os.fsencode(p) for p in paths])
        lib = _lib.load()
        _lib.check(lib.srf_feeder_read_example(arr, S1, T, int(rand_start), int(self.augment), int(self.normalize_audio),
                                               wave.data_ptr(), ln.data_ptr(), st.data_ptr()), "srf_feeder_read_example")
        dev = torch.device("cuda", torch.cuda.current_device())
        raw, nd, sd = wave.to(dev), ln.to(dev), st.to(dev)
        mix = torch.empty((1, T), dtype=torch.float32, device=dev)
        src = torch.empty((1, S1 - 1, T), dtype=torch.float32, device=dev)
        _lib.check(lib.srf_feeder_normalize(raw.data_ptr(), nd.data_ptr(), sd.data_ptr(), 1, S1, T, int(self.normalize_audio),
                                            C.c_float(EPS), mix.data_ptr(), src.data_ptr(), _lib.current_stream(dev)),
                   "srf_feeder_normalize")
        return mix[0].cpu(), src[0].cpu()

    def _getitem_host(self, idx, rand_start=0):
        """Dataset[i] without a GPU: files through the native reader, then the reference's recipe in its order
        (wham.py:183-217): the mixture is cropped only when augmenting a longer file (:183-186; else normalised over the
        WHOLE file and truncated afterwards), every source is read as [rand_start, rand_start + T) (:201), each stream is
        normalised over its own samples and padded (:189-191, :205-207), then everything is re-normalised with the
        POPULATION std of the padded mixture (:211-215)."""
        paths, T = self.paths_of(idx), self.time_samples
        crop = self.augment and self.file_frames[idx] > T
        mix = torch.from_numpy(wav_read(paths[0], rand_start, T) if crop else wav_read(paths[0]))
        if self.normalize_audio:
            mix = normalize_tensor_wav(mix)
        mix = self.safe_pad(mix)
        srcs = []
        for p in paths[1:]:
            s = torch.from_numpy(wav_read(p, rand_start, T))
            if self.normalize_audio:
                s = normalize_tensor_wav(s)
            srcs.append(self.safe_pad(s))
        if self.normalize_audio:
            mix_std = mix.numpy().std()
            mix = normalize_tensor_wav(mix, std=mix_std)
            srcs = [normalize_tensor_wav(s, std=mix_std) for s in srcs]
        return mix, torch.stack(srcs, dim=0)

    def get_generator(self, batch_size=4, shuffle=True, num_workers=4, device=None, prefetch=3, seed=0, drop_last=True,
                      rank=None, world_size=None):
        """The reference's DataLoader(batch_size, shuffle, num_workers, drop_last=True) (wham.py:219-224) as a native
        feeder: iterating it yields (mixtures [B, T], sources [B, S, T]) on `device` (default: the current MI355X).
        rank / world_size (default: torch.distributed's, when initialised; else 0 / 1): one process per GPU -- every rank
        gets its own disjoint `batch_size` examples of each global batch of batch_size * world_size (BatchFeeder)."""
        return BatchFeeder(self, batch_size, shuffle, num_workers, device, prefetch, seed, drop_last, rank=rank,
                           world_size=world_size)


class BatchFeeder:
    """Iterable over the batches of one epoch per ``iter()`` (epoch counter advances, so every epoch reshuffles / recrops).
    Pipeline per batch: native threads fill a pinned buffer -> cudaMemcpyAsync on a side stream -> srf_feeder_normalize on
    that stream -> the consumer's stream waits on the batch's event (no host synchronisation anywhere: a pinned slot goes
    back to the readers once its copy EVENT has completed, polled at the next batch).  `prefetch` batches are in flight.

    Sharding (rank, world_size): all ranks derive the same epoch order from (seed, epoch); this feeder
    delivers examples [rank * B, (rank + 1) * B) of every global batch of B * world_size -- disjoint across ranks, together
    exactly the single-process batch -- and every rank runs the same number of steps (drop_last is required)."""

    def __init__(self, dataset, batch_size, shuffle, num_workers, device, prefetch, seed, drop_last, host_only=False,
                 rank=None, world_size=None):
        if not dataset.zero_pad and any(n < dataset.time_samples for n in dataset.file_frames):
            raise _lib.SrfError("files shorter than time_samples need zero_pad=True to be batched")
        if rank is None or world_size is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            rank = (dist.get_rank() if on else 0) if rank is None else rank
            world_size = (dist.get_world_size() if on else 1) if world_size is None else world_size
        self.rank, self.world_size = int(rank), int(world_size)
        if self.world_size > 1 and not drop_last:
            raise _lib.SrfError("a sharded feeder (world size %d) needs drop_last=True: every rank must run the same number "
                                "of steps" % self.world_size)
        self.ds, self.B, self.host_only = dataset, int(batch_size), host_only
        self.S1 = 1 + len(dataset.source_names)
        self.T = dataset.time_samples
        self.prefetch = max(1, int(prefetch))
        lib = _lib.load()
        flat = [os.fsencode(p) for i in range(len(dataset)) for p in dataset.paths_of(i)]
        arr = (C.c_char_p * len(flat))(*flat)
        h = C.c_void_p()
        _lib.check(lib.srf_feeder_create_sharded(arr, len(dataset), self.S1, self.T, self.B, max(1, int(num_workers)),
                                                 int(dataset.augment), int(bool(shuffle)), int(bool(drop_last)), int(seed),
                                                 int(dataset.normalize_audio), self.rank, self.world_size, C.byref(h)),
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
        # one slot more than the prefetch depth: the slot whose copy is still leaving the host is parked, not re-armed
        n_slots = self.prefetch + (0 if host_only else 1)
        self._slots = [(mk((self.B, self.S1, self.T), torch.float32), mk((self.B, self.S1), torch.int32),
                        mk((self.B, 2), torch.float32)) for _ in range(n_slots)]

    def __len__(self):
        return int(self._lib.srf_feeder_batches_per_epoch(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.srf_feeder_destroy(h)

    def epoch_items(self):
        """Dataset indices this rank delivers in the CURRENT epoch (the one the last iter() started), in order."""
        cap = len(self.ds)
        buf = (C.c_int * cap)()
        n = self._lib.srf_feeder_epoch_items(self._h, buf, cap)
        return list(buf[:n])

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
        queued, parked = [], []             # slots being read / slots whose host -> device copy may still be in flight
        free = list(range(len(self._slots)))
        more = True

        def arm():                          # hand every free slot back to the readers (until the epoch runs out)
            nonlocal more
            while more and free and len(queued) < self.prefetch:
                slot = free.pop(0)
                if self._submit(slot):
                    queued.append(slot)
                else:
                    more = False
                    free.insert(0, slot)

        arm()
        try:
            while queued:
                slot = queued.pop(0)
                nv = C.c_int()
                _lib.check(lib.srf_feeder_wait(self._h, None, None, None, C.byref(nv)), "srf_feeder_wait")
                w, n, st = self._slots[slot]
                if self.host_only:
                    out = (w[:nv.value].clone(), n[:nv.value].clone(), st[:nv.value].clone())
                    free.append(slot)
                else:
                    out, copied = self._to_device(w, n, st, nv.value)
                    parked.append((slot, copied))
                # slots whose copy has left the host go back to the readers; only when every slot is parked (the consumer
                # is far ahead of PCIe) does the host wait -- for the oldest copy alone
                while parked and (parked[0][1].query() or not (queued or free)):
                    s0, ev = parked.pop(0)
                    ev.synchronize()
                    free.append(s0)
                arm()
                yield out
        finally:
            while queued:                   # abandoned iteration: drain what is in flight
                queued.pop(0)
                lib.srf_feeder_wait(self._h, None, None, None, None)
            for _, ev in parked:
                ev.synchronize()

    def _to_device(self, w, n, st, nv):
        """Pinned slot -> (mix, src) on the device; returns them and the event after which the slot may be overwritten.
        Stream discipline: everything the side stream writes is ALLOCATED on the side stream, so the caching
        allocator never hands it a block whose previous owner's consumer-stream kernels (the train step of an earlier
        batch) may still be queued; the outputs are then handed to the consumer's stream with wait_event + record_stream,
        so the consumer may drop them at any time without the side stream re-using them too early."""
        dev, lib = self.device, self._lib
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev), torch.cuda.stream(self._stream):
            mix = torch.empty((self.B, self.T), dtype=torch.float32, device=dev)
            src = torch.empty((self.B, self.S1 - 1, self.T), dtype=torch.float32, device=dev)
            raw = w.to(dev, non_blocking=True)
            nd = n.to(dev, non_blocking=True)
            sd = st.to(dev, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self._stream)      # the pinned slot has left the host once this completes
            _lib.check(lib.srf_feeder_normalize(raw.data_ptr(), nd.data_ptr(), sd.data_ptr(), self.B, self.S1, self.T,
                                                int(self.ds.normalize_audio), C.c_float(EPS), mix.data_ptr(), src.data_ptr(),
                                                _lib.current_stream(dev)), "srf_feeder_normalize")
            done = torch.cuda.Event()
            done.record(self._stream)
        cur.wait_event(done)                 # consumer-stream work queued after this sees the finished batch
        mix.record_stream(cur)
        src.record_stream(cur)
        return (mix[:nv], src[:nv]), copied
