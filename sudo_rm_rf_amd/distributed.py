"""Batch-sharded multi-GPU execution of the forward path.

Every reduction on the path (GlobLN, TAC group mean) is inside one example, so the path shards over
ranks by batch with NO data-path collective: one process per GPU (torch.distributed.run), each rank runs
its own examples.  The reference's only parallelism is single-process `torch.nn.DataParallel`
(run_improved_sudormrf.py:118), which scatters the batch the same way.  `torch.distributed` (backend
"nccl" = RCCL on ROCm, "gloo" on CPU for tests) is used for rendezvous, barriers, the max-over-ranks
timing reduction and the optional output all-gather only.
"""
import os

import torch
import torch.distributed as dist


def world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process -> (0, 1, 0))."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_from_env(backend=None):
    """Initialise the default process group when WORLD_SIZE > 1.  Returns (rank, world_size, device)."""
    rank, ws, local = world()
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(local)
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if use_cuda else "gloo")
        kw = {"device_id": device} if (use_cuda and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=ws, **kw)
    return rank, ws, device


def shard_slice(global_batch, rank, world_size):
    """Contiguous equal shard [lo, hi) of the batch for `rank` (the reference's DataParallel scatter on
    dim 0).  Equal shards are required so that weak-scaling numbers compare like with like."""
    if global_batch % world_size:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def barrier(device=None):
    """Device-synchronising barrier used to bracket timed regions."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(seconds, device=None):
    """The slowest rank's time (what a whole-job throughput must be computed from)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device=None):
    """Every rank's scalar as a list (rank order) on every rank -- bench.py's per-rank step times."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    ws = dist.get_world_size()
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    parts = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def separate_sharded(model, mixtures, gather=False):
    """Run `model` on this rank's shard of `mixtures` [global_batch, ch, T] (every rank passes the same
    global tensor or at least its own rows).  Returns the local estimates, or with gather=True the
    all-gathered [global_batch, S, T] tensor (the only collective; not needed for throughput)."""
    rank, ws, _ = world()
    if not dist.is_initialized() or ws == 1:
        return model(mixtures)
    lo, hi = shard_slice(mixtures.shape[0], rank, ws)
    local = model(mixtures[lo:hi])
    if not gather:
        return local
    parts = [torch.empty_like(local) for _ in range(ws)]
    dist.all_gather(parts, local.contiguous())
    return torch.cat(parts, 0)


def flat_gradient_view(params):
    """If the .grad tensors of `params` are consecutive views of ONE buffer -- what the HIP training step hands to autograd
    (engine._TrainStep.backward: srf_backward writes one flat fp32 gradient; the per-parameter views are adopted by
    autograd without a copy) -- return that buffer as a 1-D tensor covering exactly those gradients; else None."""
    grads = [p.grad for p in params]
    if not grads or any(g is None or not g.is_contiguous() or g.dtype != grads[0].dtype or g.device != grads[0].device
                        for g in grads):
        return None
    base = grads[0].untyped_storage().data_ptr()
    item = grads[0].element_size()
    pos = grads[0].data_ptr()
    for g in grads:
        if g.untyped_storage().data_ptr() != base or g.data_ptr() != pos:
            return None
        pos += g.numel() * item
    total = (pos - grads[0].data_ptr()) // item
    return torch.empty(0, dtype=grads[0].dtype, device=grads[0].device).set_(
        grads[0].untyped_storage(), grads[0].storage_offset(), (total,), (1,))


def clamp_global_mean(local_loss, min=-30.0, max=30.0):
    """The reference runner clamps the BATCH-mean loss (run_improved_sudormrf.py:169-171: `torch.clamp(l, min=-30., max=+30.)` on
    the mean over the whole DataParallel batch, losses/sisdr.py:307).  Under batch sharding every rank only has its shard's mean
    l_r; clamping that (what round 3 did) differs from the reference whenever one shard saturates and the batch mean does not (or
    the other way round): the clamp gates the gradient of a whole shard.  This is the exact form: one all-reduce of the SCALAR
    loss gives L = mean_r l_r (equal shards); the returned tensor has the value clamp(L) and the gradient
    d/d l_r = [min <= L <= max] (torch.clamp's own mask, inclusive) -- so after `allreduce_gradients` (sum over ranks / world) every
    rank holds exactly the gradient of clamp(mean of the full batch), saturated or not.  The gate stays on the device: no host
    synchronisation.  World size 1: identical to torch.clamp(local_loss, min, max)."""
    _, ws, _ = world()
    if not (dist.is_initialized() and ws > 1):
        return torch.clamp(local_loss, min=min, max=max)
    with torch.no_grad():
        g = local_loss.detach().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        g = g / ws
        gate = ((g >= min) & (g <= max)).to(local_loss.dtype)
        value = torch.clamp(g, min=min, max=max)
    return local_loss * gate + (value - local_loss.detach() * gate)


def allreduce_gradients(parameters, average=True):
    """The training step's only collective: ONE all-reduce of the flat fp32 gradient over
    RCCL / xGMI after backward, then 1/world scaling -- what replaces the reference's DataParallel gather of
    replica gradients onto GPU 0 (run_improved_sudormrf.py:118).  Every rank then runs the identical
    clip_grad_norm_ + Adam step on identical gradients, so the replicas stay bit-identical without a broadcast.
    With equal shard sizes the averaged gradient equals the gradient of the reference's batch-mean loss
    (losses/sisdr.py:307); for the runner's +-30 clamp of that batch mean use `clamp_global_mean` on the shard loss (a plain
    torch.clamp there would gate each shard by its own mean: different from the reference when a shard saturates).

    IN PLACE when the gradients already are views of one flat buffer (the HIP training step's are): one collective on
    that buffer, one scaling kernel, no concatenation and no copy back (round 2 did `torch.cat` + ~1 230 `copy_`
    launches, ~6 ms of a 133 ms cfg-4 step).  Gradients from elsewhere (plain autograd) take the gather / scatter
    path.  Returns the reduced flat gradient (aliasing the .grad tensors in the in-place case)."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return None
    _, ws, _ = world()
    active = dist.is_initialized() and ws > 1
    flat = flat_gradient_view(params)
    if flat is not None:
        if active:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            if average:
                flat.mul_(1.0 / ws)
        return flat
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    if active:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.mul_(1.0 / ws)
        off = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return flat


def time_allreduce(flat, repeats=10, device=None):
    """Milliseconds per all-reduce of `flat` (isolated: barrier, `repeats` back-to-back collectives, max over ranks) and the
    bus bandwidth it corresponds to, 2 (N-1)/N * bytes / time -- the per-link figure of a ring all-reduce (xGMI: ~153 GB/s
    per link, 7 links per GPU).  {"bytes", "ms", "bus_GBps", "world"}; world 1: zeros."""
    _, ws, _ = world()
    nbytes = flat.numel() * flat.element_size()
    if not (dist.is_initialized() and ws > 1):
        return {"bytes": nbytes, "ms": 0.0, "bus_GBps": 0.0, "world": 1}
    import time
    scratch = flat.clone()
    dev = device if device is not None else (flat.device if flat.is_cuda else None)
    dist.all_reduce(scratch)                     # warm-up (communicator set-up)
    barrier(dev)
    t0 = time.perf_counter()
    for _ in range(repeats):
        dist.all_reduce(scratch)
        scratch.mul_(1.0 / ws)                   # (keeps the values bounded; part of what a step does anyway)
    if flat.is_cuda:
        torch.cuda.synchronize(flat.device)
    sec = max_over_ranks(time.perf_counter() - t0, dev) / repeats
    return {"bytes": nbytes, "ms": 1e3 * sec, "bus_GBps": 2.0 * (ws - 1) / ws * nbytes / sec / 1e9, "world": ws}
