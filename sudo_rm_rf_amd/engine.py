"""Host-side engine: plan / workspace cache and the single srf_forward call per model forward.

Python owns tensors and module structure only; all arithmetic happens in libsudormrf_hip.so.
"""
import ctypes as C
import os
import threading
import warnings
from collections import OrderedDict

import torch

from . import _lib

_MAX_PLANS = 10                    # per device (an auto-tuned split forward holds up to 7: whole, halves, 5:3, 9:7)
# total bytes of cached workspaces per device: variable-length inference meets a new T on every call, and a cfg-2
# workspace is 1.6 GB -- the least recently used plans go first once the cap is exceeded (the plan in use never does)
_MAX_WORKSPACE_BYTES = int(os.environ.get("SRF_MAX_WORKSPACE_GB", "24")) << 30
# Inference batches are split over two HIP streams when that measures faster (see ModelEngine._splits):
#   SRF_STREAM_SPLIT = auto (default) | off | half | 5:3
_SPLIT_MODE = os.environ.get("SRF_STREAM_SPLIT", "auto")
_SPLIT_MIN_BATCH = 8
_TUNE_AFTER = int(os.environ.get("SRF_SPLIT_TUNE_AFTER", "3"))   # calls of one (batch, T) before the split auto-tune runs
# HIP-graph replay of small forwards (opt-in): a (batch, T) that keeps coming back is captured once -- srf_forward is
# capturable by construction: immutable plan, caller-owned buffers, no synchronisation, no allocation, kernel nodes only --
# and replayed.  SRF_GRAPH = off (default) | auto | always.  Measured round 2 on cfg 1 (batch 1, ~47 launches): 0.796 ms
# replayed vs 0.784 ms eager: the small forward is bound by its under-filled kernels (25-100 workgroups on 256 CUs, a serial
# k-loop per GEMM tile), not by launch gaps, so eager stays the default.
_GRAPH_MODE = os.environ.get("SRF_GRAPH", "off")
_GRAPH_AFTER = 3                    # calls of one (batch, T) before it is captured
_GRAPH_MAX_FRAMES = 4 * 3200        # batch * frames up to which a forward counts as launch-bound
_MAX_GRAPHS = 4


def _config_struct(variant, in_audio_channels, out_channels, in_channels, num_blocks, upsampling_depth,
                   enc_kernel_size, enc_num_basis, num_sources, group_size):
    return _lib.srf_config(
        variant=_lib.VARIANT_GROUPCOMM if variant == "groupcomm" else _lib.VARIANT_IMPROVED,
        in_audio_channels=in_audio_channels, out_channels=out_channels, in_channels=in_channels,
        num_blocks=num_blocks, upsampling_depth=upsampling_depth, enc_kernel_size=enc_kernel_size,
        enc_num_basis=enc_num_basis, num_sources=num_sources, group_size=group_size)


class Plan:
    """An immutable srf_plan plus the workspace it needs (owned by torch's allocator)."""

    def __init__(self, cfg_tuple, batch, T, device):
        lib = _lib.load()
        self.cfg_tuple, self.batch, self.T, self.device = cfg_tuple, batch, T, device
        cfg = _config_struct(*cfg_tuple)
        handle = C.c_void_p()
        _lib.check(lib.srf_plan_create(C.byref(cfg), batch, T, C.byref(handle)), "srf_plan_create")
        self.handle = handle
        self.workspace_bytes = lib.srf_plan_workspace_bytes(handle)
        self.num_params = lib.srf_plan_num_params(handle)
        self.frames = lib.srf_plan_frames(handle)
        self.padded_length = lib.srf_plan_padded_length(handle)
        self.num_launches = lib.srf_plan_num_launches(handle)
        self._workspace = None          # allocated at first use: a plan that is only asked for its geometry costs no HBM

    @property
    def workspace(self):
        if self._workspace is None:
            self._workspace = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)
            if self._workspace.data_ptr() % 256:
                raise _lib.SrfError("torch returned a workspace that is not 256-byte aligned")
        return self._workspace

    @property
    def held_bytes(self):
        """Device memory this plan currently holds (0 until its first forward)."""
        return self.workspace_bytes if self._workspace is not None else 0

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().srf_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def train_sizes(self):
        """(saved bytes, scratch bytes) of the training step (srf_forward_train / srf_backward)."""
        lib = _lib.load()
        return lib.srf_train_saved_bytes(self.handle), lib.srf_train_scratch_bytes(self.handle)

    def train_scratch(self):
        """Backward scratch, shared by every training step of this plan (stream-ordered reuse)."""
        if getattr(self, "_train_scratch", None) is None:
            self._train_scratch = torch.empty(self.train_sizes()[1], dtype=torch.uint8, device=self.device)
        return self._train_scratch

    def forward(self, param_ptrs, wav, out):
        lib = _lib.load()
        rc = lib.srf_forward(self.handle, param_ptrs, self.num_params, _lib.ptr(wav), _lib.ptr(out),
                             _lib.ptr(self.workspace), self.workspace_bytes,
                             _lib.current_stream(wav.device))
        _lib.check(rc, "srf_forward")

    def separate(self, param_ptrs, wav, out, stats, mixture_consistency):
        """srf_separate: statistics of the raw mixture, normalise-on-load, forward, rescale (+ mixture consistency)."""
        lib = _lib.load()
        rc = lib.srf_separate(self.handle, param_ptrs, self.num_params, _lib.ptr(wav), _lib.ptr(out), _lib.ptr(stats),
                              int(bool(mixture_consistency)), _lib.ptr(self.workspace), self.workspace_bytes,
                              _lib.current_stream(wav.device))
        _lib.check(rc, "srf_separate")

    def debug_fetch(self, what, shape):
        dst = torch.empty(shape, dtype=torch.float32, device=self.device)
        rc = _lib.load().srf_debug_fetch(self.handle, _lib.ptr(self.workspace), what, _lib.ptr(dst),
                                         dst.numel(), _lib.current_stream(self.device))
        _lib.check(rc, "srf_debug_fetch")
        return dst


class _TrainStep(torch.autograd.Function):
    """SuDORMRF.forward under autograd: srf_forward_train keeps the activations the backward needs in one
    `saved` buffer, srf_backward turns d loss / d output into all parameter gradients (one flat buffer, returned
    as per-parameter views).  Replaces torch autograd over the reference's ~1.8 k ATen nodes
    (run_improved_sudormrf.py:167-172).  A mixture that requires grad also gets its gradient (srf_backward_wav: the encoder's
    transposed convolution of the encoder-output gradient), as the reference's autograd would return it."""

    @staticmethod
    def forward(ctx, engine, out_ch, wav, *params):
        lib = _lib.load()
        x = wav.detach().to(torch.float32).contiguous()
        batch, _, T = x.shape
        dev = x.device
        with torch.cuda.device(dev), engine._run_lock(dev):
            plan = engine.plan_for(batch, T, dev)
            saved_bytes, scratch_bytes = plan.train_sizes()
            saved = torch.empty(saved_bytes, dtype=torch.uint8, device=dev)
            scratch = plan.train_scratch()
            out = torch.empty((batch, out_ch, T), dtype=torch.float32, device=dev)
            table = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
            rc = lib.srf_forward_train(plan.handle, table, len(params), _lib.ptr(x), _lib.ptr(out), _lib.ptr(saved),
                                       saved_bytes, _lib.ptr(scratch), scratch_bytes, _lib.current_stream(dev))
            _lib.check(rc, "srf_forward_train")
        ctx.plan, ctx.saved_buf, ctx.x, ctx.engine = plan, saved, x, engine
        ctx.wav_dtype = wav.dtype
        ctx.save_for_backward(*params)
        engine.last_plan = plan
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        params = ctx.saved_tensors
        plan, saved, x = ctx.plan, ctx.saved_buf, ctx.x
        dev = x.device
        g = grad_out.detach().to(torch.float32).contiguous()
        sizes = [p.numel() for p in params]
        flat = ctx.engine._flat_grad_buffer(params, sum(sizes), dev)
        grads, off = [], 0
        for p, n in zip(params, sizes):
            grads.append(flat[off:off + n].view_as(p))
            off += n
        with torch.cuda.device(dev), ctx.engine._run_lock(dev):
            scratch = plan.train_scratch()
            ptab = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
            gtab = (C.c_void_p * len(params))(*[t.data_ptr() for t in grads])
            gwav = None
            if ctx.needs_input_grad[2]:
                gwav = torch.empty_like(x)
                rc = lib.srf_backward_wav(plan.handle, ptab, gtab, len(params), _lib.ptr(x), _lib.ptr(g), _lib.ptr(saved),
                                          saved.numel(), _lib.ptr(scratch), scratch.numel(), _lib.ptr(gwav),
                                          _lib.current_stream(dev))
            else:
                rc = lib.srf_backward(plan.handle, ptab, gtab, len(params), _lib.ptr(x), _lib.ptr(g), _lib.ptr(saved),
                                      saved.numel(), _lib.ptr(scratch), scratch.numel(), _lib.current_stream(dev))
            _lib.check(rc, "srf_backward")
        ctx.saved_buf = None
        ctx.engine.last_flat_grad = flat      # (autograd adopts the views below as the .grad tensors: this IS the gradient)
        if gwav is not None and gwav.dtype != ctx.wav_dtype:
            gwav = gwav.to(ctx.wav_dtype)
        return (None, None, gwav) + tuple(grads)


def _weights(module):
    """The module's weight tensors in state_dict() order.  For an ordinary module that is exactly
    state_dict(keep_vars=True).values(); a torch.nn.DataParallel replica keeps its (broadcast, non-leaf) weights in
    `_former_parameters` of every sub-module and reports an EMPTY state_dict (torch/nn/parallel/replicate.py), so the
    tree is walked in the same pre-order state_dict uses (run_improved_sudormrf.py:118 trains through such replicas)."""
    if not getattr(module, "_is_replica", False):
        return list(module.state_dict(keep_vars=True).values())
    out = []
    for m in module.modules():
        # (a replica's _parameters only keeps the None placeholders, e.g. the bias of a bias-free conv)
        out.extend(v for v in getattr(m, "_former_parameters", {}).values() if v is not None)
    return out


class ModelEngine:
    """Per-model state: plan cache (per device / batch / length) and the parameter pointer table."""

    def __init__(self, cfg_tuple):
        self.cfg_tuple = cfg_tuple
        self._plans = OrderedDict()
        self._ptr_cache = {}
        self._lock = threading.Lock()
        self._warned_grad = False
        self.last_plan = None
        # Batch split over two streams (see _forward_split), both models.  (Round 1 had it off for GroupComm: back-to-back
        # split forwards gave a few wrong examples.  Cause, round 2: the TAC kernel's packed bias adds hit a gfx950
        # erratum -- v_pk_add_f32 with op_sel = 1 on src1 is wrong in lanes 48..63 next to another wavefront's bf16
        # MFMA, i.e. next to the other stream's GEMM -- csrc/srf_tac.hip, build.py's ISA lint,
        # tests/test_gpu_model.py::test_split_forward_stress.)  bench.py switches the split off for its per-kernel
        # profiling pass.
        self.multi_stream = True
        self._side_streams = {}
        self._split_choice = {}
        self._seen = {}
        self._graphs = OrderedDict()
        self._graph_seen = {}
        self._run_locks = {}
        self.last_flat_grad = None

    def _flat_grad_buffer(self, params, total, device):
        """The flat fp32 buffer srf_backward writes all parameter gradients into (zeroed: the weight-gradient kernels
        accumulate).  A fresh block from torch's caching allocator per step (allocation = a free-list pop, the fill = one
        memset kernel): autograd ADOPTS the per-parameter views as .grad tensors or keeps them alive inside the graph (a
        DataParallel replica's gradients sit in its Broadcast node until all replicas are done; gradient accumulation adds
        later steps INTO the first step's views), so the lifetime of a step's buffer is not the engine's to decide -- a
        persistent buffer re-zeroed per step corrupted exactly those two cases (tests/test_gpu_train.py)."""
        return torch.zeros(total, dtype=torch.float32, device=device)

    def _run_lock(self, device):
        """One re-entrant lock per device: a forward is a chain of dependent launches into a workspace, so the launches
        of two Python threads calling the same module on the same GPU must not interleave (the reference's forward
        is re-entrant); threads on different GPUs (DataParallel replicas share this object) do not
        contend."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with self._lock:
            lk = self._run_locks.get(idx)
            if lk is None:
                lk = self._run_locks[idx] = threading.RLock()
        return lk

    def plan_for(self, batch, T, device, lane=0, keep=()):
        """lane: plans of different stream lanes never share a workspace, even for equal sub-batch sizes.  Lane 0 (work
        launched on the caller's current stream) is additionally keyed by that stream: callers on different streams
        get different workspaces.  keep: cache keys (Plan.cache_key) of other plans the CURRENT forward is using -- the
        sibling lanes of a split forward -- which the eviction this call may trigger must leave alone."""
        if lane == 0:
            lane = (0, torch.cuda.current_stream(device).cuda_stream)
        key = (device.index if device.index is not None else torch.cuda.current_device(), batch, T, lane)
        with self._lock:
            plan = self._plans.get(key)
            if plan is None:
                plan = Plan(self.cfg_tuple, batch, T, device)
                plan.cache_key = key
                self._plans[key] = plan
            else:
                self._plans.move_to_end(key)
            self._evict(key[0], keep={key, *keep}, incoming=plan.workspace_bytes - plan.held_bytes)
        return plan

    def _evict(self, dev_index, keep, incoming=0):
        """LRU per device: at most _MAX_PLANS plans and _MAX_WORKSPACE_BYTES of workspaces (the reference allocates and
        frees its activations on every call; a cached workspace is the same memory held a little longer).  A dropped
        workspace goes back to torch's caching allocator, whose stream-ordered reuse keeps in-flight kernels safe.
        `keep`: keys that stay whatever the caps say (every plan of the forward in flight: for big shapes -- cfg 5: 11.8 GB
        whole, 2 x 6 GB halves -- the lanes of a split would otherwise evict each other on every call); `incoming`: bytes
        the plan being handed out is about to allocate."""
        mine = [k for k in self._plans if k[0] == dev_index]
        total = incoming + sum(self._plans[k].held_bytes for k in mine)
        for k in mine:                                   # oldest first
            over_count = len(mine) > _MAX_PLANS
            if not over_count and total <= _MAX_WORKSPACE_BYTES:
                break
            if k in keep or (not over_count and self._plans[k].held_bytes == 0):
                continue                                 # (a plan without a workspace frees nothing)
            total -= self._plans[k].held_bytes
            del self._plans[k]
            mine = [m for m in mine if m != k]

    def _param_table(self, params, device):
        key = tuple(p.data_ptr() for p in params)
        dkey = device.index
        cached = self._ptr_cache.get(dkey)
        if cached is not None and cached[0] == key:
            return cached[1]
        arr = (C.c_void_p * len(key))(*key)
        self._ptr_cache[dkey] = (key, arr)
        return arr

    def _run_train(self, module, wav, expected_channels):
        params = _weights(module)
        for p in params:
            if p.device != wav.device or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.SrfError("all parameters must be contiguous float32 on %s" % wav.device)
        if wav.shape[0] == 0 or wav.shape[-1] == 0:
            raise RuntimeError("empty input %s" % (tuple(wav.shape),))
        return _TrainStep.apply(self, module.num_sources * expected_channels, wav, *params)

    def run(self, module, wav, expected_channels):
        if not isinstance(wav, torch.Tensor):
            raise TypeError("input must be a torch.Tensor")
        if wav.dim() != 3:
            # the reference fails on non-3D input too (Conv1d on the padded tensor)
            raise RuntimeError("expected input of shape [batch, %d, time], got %s" %
                               (expected_channels, tuple(wav.shape)))
        if wav.shape[1] != expected_channels:
            raise RuntimeError("expected %d input channel(s), got %d" % (expected_channels, wav.shape[1]))
        if wav.device.type != "cuda":
            raise _lib.SrfError(
                "sudo_rm_rf_amd runs on an MI355X only: input is on %s.  There is deliberately no CPU "
                "fallback (use the reference implementation for CPU inference)." % wav.device)
        # (from the state_dict tensors, not module.parameters(): DataParallel replicas hold their weights as plain
        # tensors -- views that require grad through the Broadcast node -- and report no parameters at all)
        weights = _weights(module)
        wants_grad = torch.is_grad_enabled() and any(t.requires_grad for t in weights)
        # model.train() + grad mode = the training step (what the reference's runner does before its loop,
        # run_improved_sudormrf.py:144); model.eval() always takes the fused inference path
        if wants_grad and module.training:
            return self._run_train(module, wav, expected_channels)
        if wants_grad and not self._warned_grad:
            self._warned_grad = True
            warnings.warn("sudo_rm_rf_amd: forward in eval() mode with gradients enabled returns a DETACHED output (the "
                          "fused inference path keeps no activations); call model.train() for the HIP training step or "
                          "wrap inference in torch.no_grad()", stacklevel=3)
        params = [p.detach() for p in weights]
        for p in params:
            if p.device != wav.device or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.SrfError("all parameters must be contiguous float32 on %s" % wav.device)
        # the reference's pad buffer is float32 whatever the input dtype (improved_sudormrf.py:312)
        x = wav.detach().to(torch.float32).contiguous()
        batch, _, T = x.shape
        if batch == 0 or T == 0:
            raise RuntimeError("empty input %s" % (tuple(wav.shape),))
        with torch.cuda.device(x.device), self._run_lock(x.device):
            plan = self.plan_for(batch, T, x.device)
            if plan.num_params != len(params):
                raise _lib.SrfError("state_dict has %d tensors, plan expects %d" %
                                    (len(params), plan.num_params))
            out_ch = module.num_sources * expected_channels
            table = self._param_table(params, x.device)
            if self._wants_graph(plan, batch, T, x.device):
                out = self._forward_graph(batch, T, out_ch, x, params, table)
                self.last_plan = plan
                return out
            out = torch.empty((batch, out_ch, T), dtype=torch.float32, device=x.device)
            self._forward_split(self._splits(batch, T, x, out, table), x, out, table)
            self.last_plan = plan
        return out

    def separate(self, module, mixture, mixture_consistency):
        """The reference's caller-side recipe around model() (README.md:100-114) as one srf_separate call: mixture
        [batch, 1, time] RAW (un-normalised) on the MI355X -> estimates [batch, num_sources, time] in the mixture's scale."""
        if mixture.device.type != "cuda":
            raise _lib.SrfError("sudo_rm_rf_amd runs on an MI355X only: input is on %s" % mixture.device)
        params = [p.detach() for p in _weights(module)]
        for p in params:
            if p.device != mixture.device or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.SrfError("all parameters must be contiguous float32 on %s" % mixture.device)
        # same shape contract as run(): srf_separate reads batch * T floats, so a [B, 2, T] tensor handed to a 1-channel
        # model would silently be processed as its first B * T samples
        channels = int(getattr(module, "in_audio_channels", 1) or 1)
        if mixture.dim() != 3:
            raise RuntimeError("expected a mixture of shape [batch, %d, time], got %s" % (channels, tuple(mixture.shape)))
        if mixture.shape[1] != channels or channels != 1:
            raise _lib.SrfError("separate() takes single-channel mixtures [batch, 1, time] (the README recipe); got %s for a "
                                "model with %d input channel(s)" % (tuple(mixture.shape), channels))
        x = mixture.detach().to(torch.float32).contiguous()
        batch, _, T = x.shape
        if batch == 0 or T == 0:
            raise RuntimeError("empty input %s" % (tuple(mixture.shape),))
        with torch.cuda.device(x.device), self._run_lock(x.device):
            plan = self.plan_for(batch, T, x.device)
            if plan.num_params != len(params):
                raise _lib.SrfError("state_dict has %d tensors, plan expects %d" % (len(params), plan.num_params))
            out = torch.empty((batch, module.num_sources, T), dtype=torch.float32, device=x.device)
            stats = torch.empty((batch, 2), dtype=torch.float32, device=x.device)
            plan.separate(self._param_table(params, x.device), x, out, stats, mixture_consistency)
            self.last_plan = plan
        return out

    # ---- HIP-graph replay of small forwards -----------------------------------------------------------
    def _wants_graph(self, plan, batch, T, device):
        if _GRAPH_MODE == "off" or torch.cuda.is_current_stream_capturing():
            return False
        if _GRAPH_MODE != "always" and batch * plan.frames > _GRAPH_MAX_FRAMES:
            return False
        key = (device.index, batch, T)
        n = self._graph_seen.get(key, 0) + 1
        self._graph_seen[key] = n
        if len(self._graph_seen) > 4096:
            self._graph_seen.clear()
        return _GRAPH_MODE == "always" or n >= _GRAPH_AFTER

    def _forward_graph(self, batch, T, out_ch, x, params, table):
        """Replay (capturing on first use) the forward of this (device, batch, T, weight tensors) as ONE graph launch.  The
        graph owns a plan + workspace of its own and static input / output buffers: the call copies the input in, replays,
        and returns a copy of the output (two small copies; the forwards this is used for are tiny)."""
        dev = x.device
        key = (dev.index, batch, T, tuple(p.data_ptr() for p in params))
        ent = self._graphs.get(key)
        if ent is None:
            plan = Plan(self.cfg_tuple, batch, T, dev)
            s_in = x.clone()
            s_out = torch.empty((batch, out_ch, T), dtype=torch.float32, device=dev)
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):          # warm-up off the capture: one-off host work (kernel attributes, caches)
                plan.forward(table, s_in, s_out)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                plan.forward(table, s_in, s_out)
            ent = (graph, plan, s_in, s_out, table)      # (table: the ctypes pointer array the capture read must stay alive)
            self._graphs[key] = ent
            while len(self._graphs) > _MAX_GRAPHS:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        graph, _, s_in, s_out, _ = ent
        s_in.copy_(x)
        graph.replay()
        return s_out.clone()

    # ---- batch split over two streams ---------------------------------------------------------------
    # Examples are independent, so a batch may run as two sub-batches on two HIP streams: the tail
    # of one stream's kernel (partial last round of GEMM tiles, the 32-block finalize kernels, epilogue write
    # bursts) is filled with the other's blocks.  Measured on one MI355X: cfg 2 batch 32 8.20 -> 7.72 ms (16+16),
    # cfg 3 4.97 -> 4.73 ms (20+12), cfg 5 65.2 -> 63.7 ms (10+6), cfg 4 29.5 -> 29.3 ms (20+12; 16+16 loses 3 %).
    # Which split wins depends on how the tile counts quantise, so the first call of a (batch, length) times the
    # candidates (2 forwards each, all produce the correct output) and keeps the fastest.
    def _split_candidates(self, batch):
        if _SPLIT_MODE == "off" or batch < _SPLIT_MIN_BATCH or not self.multi_stream:
            return [(batch,)]
        half = (batch - batch // 2, batch // 2)
        skew = (batch - (3 * batch) // 8, (3 * batch) // 8)
        mild = (batch - (7 * batch) // 16, (7 * batch) // 16)       # 9 : 7 (round 5: best or equal-best on cfgs 2 and 3,
        #                                                             profiles/r05_stream_split_sweep.txt)
        if _SPLIT_MODE == "half":
            return [half]
        if ":" in _SPLIT_MODE:          # explicit weights, e.g. "5:3" or "1:1:1" (experiments)
            wts = [int(v) for v in _SPLIT_MODE.split(":")]
            parts, left = [], batch
            for i, wv in enumerate(wts):
                n = left if i == len(wts) - 1 else max(1, round(batch * wv / sum(wts)))
                parts.append(min(n, left - (len(wts) - 1 - i)))
                left -= parts[-1]
            return [tuple(parts)]
        out = [(batch,), half]
        for c in (skew, mild):
            if c not in out and min(c) > 0:
                out.append(c)
        return out

    def _forward_split(self, parts, x, out, table):
        dev = x.device
        if len(parts) == 1:
            self.plan_for(parts[0], x.shape[-1], dev).forward(table, x, out)
            return
        cur = torch.cuda.current_stream(dev)
        streams = self._side_streams.setdefault(dev.index, [])
        while len(streams) < len(parts):
            streams.append(torch.cuda.Stream(dev))
        lo, held = 0, []
        for lane, n in enumerate(parts):
            st = streams[lane]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                plan = self.plan_for(n, x.shape[-1], dev, lane=lane + 1, keep=held)
                held.append(plan.cache_key)
                plan.forward(table, x[lo:lo + n], out[lo:lo + n])
            x.record_stream(st)
            out.record_stream(st)
            lo += n
        for st in streams[:len(parts)]:
            cur.wait_stream(st)

    def _splits(self, batch, T, x, out, table):
        if not self.multi_stream:
            return (batch,)
        key = (x.device.index, batch, T)
        choice = self._split_choice.get(key)
        if choice is not None:
            return choice
        cands = self._split_candidates(batch)
        if len(cands) > 1 and _SPLIT_MODE == "auto":
            # The tune costs 20 extra forwards, a host sync and up to 6 more workspaces: only worth it for a shape that
            # keeps coming back (a training / benchmark loop), not for variable-length inference where every call
            # brings a new T.  Until a shape has been seen _TUNE_AFTER times it runs un-split on the caller's stream.
            seen = self._seen.get(key, 0) + 1
            self._seen[key] = seen
            if len(self._seen) > 4096:
                self._seen.clear()
            if seen < _TUNE_AFTER:
                return (batch,)
        if len(cands) == 1:
            choice = cands[0]
        else:
            best = None
            for parts in cands:
                self._forward_split(parts, x, out, table)          # warm-up (plan creation, first-touch)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    self._forward_split(parts, x, out, table)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                # a split must beat the un-split forward by > 1.5 % to be chosen, another split by > 0.5 %
                if best is None or ms < best[0] * (0.985 if len(best[1]) == 1 else 0.995):
                    best = (ms, parts)
            choice = best[1]
        self._split_choice[key] = choice
        return choice
