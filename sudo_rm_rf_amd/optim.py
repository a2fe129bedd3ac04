"""Fused gradient clipping + Adam: the last two lines of the reference runner's step,
``torch.nn.utils.clip_grad_norm_(model.parameters(), clip); opt.step()`` (run_improved_sudormrf.py:172-176), as two
HIP launches over every parameter at once (csrc/srf_optim.hip), without the host sync clip_grad_norm_ needs."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

_CHECK_FINITE = os.environ.get("SRF_CHECK_FINITE") == "1"   # check_finite() after every step (one host sync per step)


class FusedClipAdam(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.Adam(params, lr, betas, eps)`` (no weight decay / amsgrad) that also applies
    ``clip_grad_norm_(params, clip_grad_norm)`` inside ``step()``.  State (``exp_avg``, ``exp_avg_sq``, ``step``) uses
    torch's names, so ``state_dict()`` is interchangeable with torch.optim.Adam's."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clip_grad_norm=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, clip_grad_norm=clip_grad_norm)
        super().__init__(params, defaults)
        self._tables = {}
        self.last_grad_norm = None          # device scalar: total gradient norm before clipping (last step)

    def _table(self, gi, group, plist):
        """Device tables of one parameter group: {param, grad, exp_avg, exp_avg_sq, numel} per tensor and the chunk list.
        The chunk list, the bucket scratch and the table's device storage depend on the tensors' SIZES only and are built
        once; the pointers are compared with the last step's on the host and, when any moved (the HIP training step hands
        out a fresh flat gradient buffer per step; load_state_dict replaces the state tensors), re-sent through a PINNED
        staging buffer with a non-blocking copy on the launch stream -- stream order puts it behind the previous step's
        kernels and ahead of this step's.  (A pageable `.to(device)` here made step() wait for the whole backward: the host
        could never run ahead, and every host-side section of the next step was GPU idle time -- 1.5-2 ms of a 32-ms step.)"""
        dev = plist[0].device
        sizes = tuple(p.numel() for p in plist)
        cached = self._tables.get(gi)
        if cached is None or cached["sizes"] != sizes or cached["dev"] != dev:
            chunk = _lib.load().srf_opt_chunk_size()
            chunks = [(i, c) for i, n in enumerate(sizes) for c in range((n + chunk - 1) // chunk)]
            cached = self._tables[gi] = {
                "sizes": sizes, "dev": dev, "desc": None,
                "tens": torch.empty((len(plist), 5), dtype=torch.int64, device=dev),
                "chs": torch.tensor(chunks, dtype=torch.int32).to(dev),
                "buckets": torch.empty(_lib.STAT_BUCKETS, dtype=torch.float64, device=dev),
                "norm": torch.empty(1, dtype=torch.float32, device=dev)}
        desc = np.empty((len(plist), 5), dtype=np.int64)
        for i, p in enumerate(plist):
            st = self.state[p]
            desc[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), sizes[i])
        # The table is overwritten IN PLACE: safe while every step's copy and kernel are ordered on one stream.  A caller that
        # moves the optimizer to another stream makes the new stream wait for the old one first.
        cur = torch.cuda.current_stream(dev)
        if cached.get("stream") is not None and cached["stream"] != cur:
            cur.wait_stream(cached["stream"])
        cached["stream"] = cur
        if cached["desc"] is None or not np.array_equal(cached["desc"], desc):
            # one persistent pinned staging buffer per group; the event says when its previous copy has left the host
            if cached.get("staging") is None:
                cached["staging"] = torch.empty((len(plist), 5), dtype=torch.int64).pin_memory()
                cached["staged"] = torch.cuda.Event()
            else:
                cached["staged"].synchronize()
            cached["staging"].copy_(torch.from_numpy(desc))
            cached["tens"].copy_(cached["staging"], non_blocking=True)
            cached["staged"].record(cur)
            cached["desc"] = desc
        return cached["tens"], cached["chs"], cached["buckets"], cached["norm"]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            dev = plist[0].device
            for p in plist:
                if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() or p.device.type != "cuda":
                    raise _lib.SrfError("FusedClipAdam: parameters must be contiguous float32 tensors on one MI355X")
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            # torch.optim.Adam's state_dict keeps `step` as a float tensor: accept it
            step = int(self.state[plist[0]]["step"]) + 1
            for p in plist:
                self.state[p]["step"] = step
            tens, chs, buckets, norm = self._table(gi, group, plist)
            b1, b2 = group["betas"]
            with torch.cuda.device(dev):
                rc = lib.srf_clip_adam_step(_lib.ptr(tens), _lib.ptr(chs), chs.shape[0], _lib.ptr(buckets),
                                            C.c_float(group["clip_grad_norm"]), C.c_float(group["lr"]), C.c_float(b1),
                                            C.c_float(b2), C.c_float(group["eps"]), step, _lib.ptr(norm),
                                            _lib.current_stream(dev))
            _lib.check(rc, "srf_clip_adam_step")
            self.last_grad_norm = norm
        if _CHECK_FINITE:
            self.check_finite()
        return loss

    def check_finite(self):
        """Host check of the last step's total gradient norm (ONE synchronising read; off the hot path unless
        SRF_CHECK_FINITE=1 asks for it after every step).  The HIP training forward runs its 1x1 convolutions on two FP16 parts
        per operand: an activation beyond fp16's range (|x| >= 65520), where the fp32 reference would still be finite, turns the
        step non-finite instead of being silently clamped -- this is where that becomes a clear error."""
        if self.last_grad_norm is not None and not bool(torch.isfinite(self.last_grad_norm).all()):
            raise _lib.SrfError(
                "FusedClipAdam: the gradient norm of the last step is not finite.  If the loss itself is finite in the fp32 "
                "reference, an operand of the training forward's GEMMs left fp16's range (|x| >= 65520): re-run with the "
                "three-bf16-part training GEMMs (sudo_rm_rf_amd.ops.set_debug_flags(16384)), which keep fp32's exponent range")

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam's state_dict as well: its param groups carry no clip_grad_norm (this optimizer's own
        value is kept) and its `step` is a float tensor (cast on use); the device pointer table is rebuilt."""
        keep = [g["clip_grad_norm"] for g in self.param_groups]
        super().load_state_dict(state_dict)
        for g, c in zip(self.param_groups, keep):
            g.setdefault("clip_grad_norm", c)
        # torch hands the donor's `step` tensors over by reference (Optimizer.load_state_dict does not clone them for
        # non-capturable optimizers): an int of our own, or the donor's next step() would advance ours as well
        for st in self.state.values():
            if "step" in st:
                st["step"] = int(st["step"])
        self._tables.clear()
