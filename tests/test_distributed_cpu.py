"""N > 1 path on CPU: world_size-2 gloo processes exercise the batch sharding, the barrier, the
max-over-ranks timing reduction and the output all-gather used by bench.py / separate_sharded."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from sudo_rm_rf_amd import distributed as D
    from oracle import torch_oracle
    from oracle.schema import ModelConfig
    from oracle.weights import make_mixture, make_state_dict
    torch.set_num_threads(1)
    r, ws, dev = D.init_from_env(backend="gloo")
    assert (r, ws) == (rank, world) and dev.type == "cpu"
    # sharding covers the batch exactly once
    lo, hi = D.shard_slice(8, r, ws)
    assert hi - lo == 4 and lo == 4 * r
    # timing reduction: slowest rank wins
    t = D.max_over_ranks(1.0 + rank)
    assert t == pytest.approx(2.0)
    D.barrier(dev)
    # a stand-in "model" (the CPU oracle): sharded + gathered == unsharded, examples are independent
    cfg = ModelConfig("improved", 16, 32, 1, 3, 21, 16, 2)
    sd = torch_oracle.to_torch(make_state_dict(cfg, 5))
    wav = torch.from_numpy(make_mixture(4, 640, 6))
    model = lambda x: torch_oracle.forward(cfg, sd, x)
    with torch.no_grad():
        full = model(wav)
        local = D.separate_sharded(model, wav)
        gathered = D.separate_sharded(model, wav, gather=True)
    assert local.shape[0] == 2
    # training step collective: sharded batch-mean gradients, all-reduced and averaged == full-batch gradient
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    tgt = torch.from_numpy(make_mixture(4, 640, 7, channels=2, normalize=False))
    lo2, hi2 = D.shard_slice(4, r, ws)
    ((torch_oracle.forward(cfg, sdg, wav[lo2:hi2]) - tgt[lo2:hi2]) ** 2).mean().backward()
    flat = D.allreduce_gradients(list(sdg.values()))
    sdf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ((torch_oracle.forward(cfg, sdf, wav) - tgt) ** 2).mean().backward()
    gerr = max(float((a.grad - b.grad).abs().max() / b.grad.abs().max().clamp_min(1e-12))
               for a, b in zip(sdg.values(), sdf.values()))
    assert gerr < 1e-4, gerr
    assert flat.numel() == sum(v.numel() for v in sd.values())
    q.put((rank, float((gathered - full).abs().max()), float((local - full[2 * rank:2 * rank + 2]).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    for _, e_gather, e_local in res:
        assert e_gather < 1e-6 and e_local < 1e-6


def _train_worker(rank, world, port, q, tree):
    """The 8-GPU training path's glue on 2 CPU ranks: rank-aware feeder, the flat gradient buffer
    all-reduced IN PLACE, bench.py's train loop (barrier, timed steps, max over ranks, all-reduce timing) with a stub model."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from sudo_rm_rf_amd import distributed as D, feeder
    torch.set_num_threads(1)
    r, ws, dev = D.init_from_env(backend="gloo")
    # ---- feeder: rank / world picked up from the process group; ranks' epochs are disjoint and cover the set
    ds = feeder.Dataset(root_dirpath=tree, task="sep_clean", split="tr", sample_rate=8000, timelength=0.2,
                        normalize_audio=True, n_samples=0, zero_pad=True, augment=True, min_or_max="min")
    bf = feeder.BatchFeeder(ds, 1, True, 2, None, 2, 9, True, host_only=True)
    assert (bf.rank, bf.world_size) == (rank, world)
    batches = list(bf)
    mine = torch.tensor(bf.epoch_items(), dtype=torch.int64)
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    allv = torch.cat(both).tolist()
    assert len(batches) == len(bf) == len(ds) // world and len(set(allv)) == len(allv) == world * len(bf)
    # ---- a stub "model" whose gradients are views of ONE flat buffer, like the HIP training step's
    torch.manual_seed(0)                                                    # same weights on every rank
    shapes = [(4, 3), (5,), (2, 2, 2)]
    params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    flat = torch.zeros(sum(p.numel() for p in params))
    state = {"calls": 0}

    def step():
        x = batches[state["calls"] % len(batches)][0].double().mean()       # this rank's data decides its gradient
        state["calls"] += 1
        flat.zero_()
        off = 0
        for i, p in enumerate(params):
            n = p.numel()
            v = flat[off:off + n].view_as(p)
            v += (i + 1) * float(x) + rank
            p.grad = v
            off += n
        red = D.allreduce_gradients(params)
        assert red.data_ptr() == flat.data_ptr() and red.numel() == flat.numel()      # in place: no cat, no copy back
        for p in params:
            assert p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
        return red.sum()

    assert D.flat_gradient_view(params) is None                             # (no gradients yet)
    out = bench.train_loop(step, lambda: flat, steps=3, warmup=1, rank=rank, world=world, dev=dev)
    assert state["calls"] == 1 + 1 + 3
    assert len(out["per_rank_ms_per_step"]) == world and out["ms_per_step"] >= max(out["per_rank_ms_per_step"]) - 1e-6
    ar = out["allreduce"]
    assert ar["world"] == world and ar["bytes"] == 4 * flat.numel() and ar["ms"] > 0 and ar["bus_GBps"] > 0
    # the averaged gradient is identical on every rank: gather and compare
    g = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(g, flat)
    assert all(torch.equal(g[0], t) for t in g)
    # gradients that are NOT one buffer take the gather / scatter path and give the same answer
    for p in params:
        p.grad = p.grad.clone() + rank
    want = [p.grad.clone() for p in params]
    red2 = D.allreduce_gradients(params)
    assert D.flat_gradient_view(params) is None and red2.numel() == flat.numel()
    for p, w0 in zip(params, want):
        assert torch.allclose(p.grad, w0 - rank + (world - 1) / 2.0)
    q.put((rank, allv, float(flat.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_training_glue(tmp_path):
    from oracle import feeder_oracle
    feeder_oracle.make_fake_wham(str(tmp_path), task="sep_clean", seed=3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]


def _clamp_worker(rank, world, port, q):
    """Exact batch-mean clamp under sharding: rank 0's shard loss saturates (> +30), the batch mean does not
    (and a second case the other way round): the sharded step must reproduce the single-process gradient of
    clamp(mean over the full batch) -- a per-shard clamp would drop (or keep) a whole shard's gradient."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sudo_rm_rf_amd import distributed as D
    torch.set_num_threads(1)
    D.init_from_env(backend="gloo")
    out = []
    # per-example "losses" = w . x_b; batch of 4, two per rank.  case A: shard means (+50, -20) -> batch mean +15 (inside);
    # case B: shard means (+50, +40) -> batch mean +45 (saturated: zero gradient everywhere); case C: (-10, +20): nothing clamps
    for shard_means in ((50.0, -20.0), (50.0, 40.0), (-10.0, 20.0)):
        w = torch.nn.Parameter(torch.tensor([1.0, 2.0, -1.0], dtype=torch.float64))
        base = torch.tensor([[1.0, 0.5, 0.25], [0.5, 1.0, 2.0], [2.0, -1.0, 0.5], [-0.5, 0.25, 1.0]], dtype=torch.float64)
        x = base.clone()
        for r in range(world):            # scale every shard so that its mean loss is the prescribed value
            cur = (x[2 * r:2 * r + 2] @ w.detach()).mean()
            x[2 * r:2 * r + 2] *= shard_means[r] / cur
        # reference: one process, the whole batch
        wf = torch.nn.Parameter(w.detach().clone())
        lf = torch.clamp((x @ wf).mean(), min=-30.0, max=30.0)
        lf.backward()
        # sharded
        ll = (x[2 * rank:2 * rank + 2] @ w).mean()
        l = D.clamp_global_mean(ll, min=-30.0, max=30.0)
        l.backward()
        D.allreduce_gradients([w])
        # what a per-shard clamp would have produced (must differ in case A)
        wp = torch.nn.Parameter(w.detach().clone())
        torch.clamp((x[2 * rank:2 * rank + 2] @ wp).mean(), min=-30.0, max=30.0).backward()
        D.allreduce_gradients([wp])
        out.append((float(l), float(lf), (w.grad - wf.grad).abs().max().item(), (wp.grad - wf.grad).abs().max().item()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_exact_batch_mean_clamp():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_clamp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, out in res:
        (la, lfa, ea, pa), (lb, lfb, eb, pb), (lc, lfc, ec, pc) = out
        assert la == pytest.approx(15.0) and lfa == pytest.approx(15.0) and ea < 1e-12      # inside: exact ...
        assert pa > 1e-3                                                                    # ... where a per-shard clamp is not
        assert lb == pytest.approx(30.0) and lfb == pytest.approx(30.0) and eb < 1e-12      # saturated batch mean: zero gradient
        assert lc == pytest.approx(5.0) and ec < 1e-12 and pc < 1e-12                       # nothing clamps: all three agree


def test_clamp_global_mean_single_process_is_torch_clamp():
    from sudo_rm_rf_amd import distributed as D
    for v in (-40.0, 3.0, 31.0):
        a = torch.tensor(v, requires_grad=True)
        b = torch.tensor(v, requires_grad=True)
        D.clamp_global_mean(a).backward()
        torch.clamp(b, min=-30.0, max=30.0).backward()
        assert float(D.clamp_global_mean(a).detach()) == float(torch.clamp(b, min=-30.0, max=30.0).detach()) and float(a.grad) == float(b.grad)


def test_shard_slice_rejects_uneven():
    from sudo_rm_rf_amd.distributed import shard_slice
    with pytest.raises(ValueError):
        shard_slice(10, 0, 4)
    assert [shard_slice(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]


def test_single_process_is_a_no_op():
    from sudo_rm_rf_amd import distributed as D
    assert D.max_over_ranks(0.25) == 0.25
    D.barrier(torch.device("cpu"))
    x = torch.randn(3, 1, 10)
    assert torch.equal(D.separate_sharded(lambda t: t * 2, x), x * 2)


def test_bench_launcher_spawns_ranks_on_cpu():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run must become the launcher itself (one rank per GPU, the
    driver's own command line): --launch-check runs that path up to the first kernel -- spawn, rendezvous (gloo: no GPU
    here), rank binding, barrier, max-over-ranks and per-rank gather -- and prints one JSON line on rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["launch_check"] and d["n_gpus"] == 2 and d["requested_gpus"] == 2
    assert d["max_over_ranks"] == 2.0 and d["per_rank"] == [0.0, 1.0]


@pytest.mark.parametrize("train", [False, True], ids=["inference", "training"])
def test_bench_end_to_end_on_cpu_stub(train):
    """`python bench.py --gpus 2 --cpu-stub [--train]`: the whole glue of a multi-GPU bench run -- launcher,
    rendezvous, the timed region (the same timed_forwards / train_loop the GPU path uses), barriers, max over ranks, per-rank
    gathers, the gradient all-reduce and its timing, the JSON assembly (the same forward_result) -- over gloo with a stand-in
    model: one JSON line on rank 0 with the driver's fields, aggregate value over both ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-stub", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd + (["--train"] if train else []), capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline"):
        assert k in d, k
    assert d["cpu_stub"] and d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["ranks"]["backend"] == "gloo"
    if train:
        assert len(d["per_rank_ms_per_step"]) == 2
        ar = d["gradient_allreduce"]
        assert ar["world"] == 2 and ar["bytes"] > 0 and ar["ms"] > 0 and ar["bus_GBps"] > 0
        # weak-scaling arithmetic: the line's value is BOTH ranks' examples over the slowest rank's step time,
        # i.e. 2 x the per-rank figure -- a first real SCALE run cannot report a per-rank number as the aggregate
        per_rank = 4 * (2000 / 8000) / (d["ms_per_step"] * 1e-3)
        assert d["value"] == pytest.approx(2 * per_rank, rel=1e-6)
        assert d["ms_per_step"] == pytest.approx(max(d["per_rank_ms_per_step"]), rel=0.25)
    else:
        assert len(d["ranks"]["ms_per_step_by_rank"]) == 2 and d["config"]["global_batch"] == 2 * 4
        # whole-job aggregate: both ranks' examples over the slowest rank's time
        assert d["value"] == pytest.approx(2 * 4 * (2000 / 8000) * 3 / (d["ms_per_step"] * 3e-3), rel=1e-6)
        assert d["roofline"]["frac"] == d["forward_roofline"]["frac"]
