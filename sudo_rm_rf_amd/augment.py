"""On-GPU online remix augmentation: the lines the reference's training loops run right
before the forward (experiments/run_improved_sudormrf.py:150-164; run_fuss_separation.py:195-215 generalises them to
n sources), as three small HIP kernels instead of ~25 ATen ops."""
import torch

from . import _lib


def online_remix(clean_wavs, eps=1e-8):
    """clean_wavs: [batch, n_sources, time] float32 on the MI355X -> (mixtures [batch, time], sources [batch,
    n_sources, time]), both normalised per row exactly like the runner's normalize_tensor_wav (:127-131).

    Random numbers: one ``torch.randperm(n_sources)`` followed by one ``torch.randperm(batch)`` per source, from
    torch's default CPU generator -- the same calls, in the same order, as the runner makes, so a seeded run draws
    the same permutations."""
    if clean_wavs.dim() != 3:
        raise RuntimeError("expected [batch, n_sources, time], got %s" % (tuple(clean_wavs.shape),))
    if clean_wavs.device.type != "cuda":
        raise _lib.SrfError("sudo_rm_rf_amd.augment runs on an MI355X only (input on %s)" % clean_wavs.device)
    B, S, T = clean_wavs.shape
    x = clean_wavs.detach().to(torch.float32).contiguous()
    dev = x.device
    src_s = torch.randperm(S)
    src_b = torch.stack([torch.randperm(B) for _ in range(S)])
    src_s_d = src_s.to(torch.int32).to(dev)
    src_b_d = src_b.to(torch.int32).to(dev).contiguous()
    lib = _lib.load()
    mix = torch.empty((B, T), dtype=torch.float32, device=dev)
    out = torch.empty_like(x)
    scratch = torch.empty(lib.srf_online_remix_scratch_bytes(B, S), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.srf_online_remix(_lib.ptr(x), _lib.ptr(src_b_d), _lib.ptr(src_s_d), B, S, T, float(eps), _lib.ptr(mix),
                                  _lib.ptr(out), _lib.ptr(scratch), _lib.current_stream(dev))
    _lib.check(rc, "srf_online_remix")
    return mix, out
