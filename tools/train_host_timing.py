#!/usr/bin/env python3
"""Where does the HOST spend a training step?  bench.py --train reports host_issue_ms_per_step; when that equals the step time the
host is not ahead of the GPU and every host-side section is GPU idle time.  This times the sections of the runner's loop body on
the host (no synchronisation added) over a few steady steps, and the same with a device synchronise after every section (= the
GPU time of each section).

    python tools/train_host_timing.py [--workload cfg2_improved_u16] [--steps 6]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2_improved_u16")
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib
    import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as gc
    import sudo_rm_rf.dnn.models.improved_sudormrf as imp
    from sudo_rm_rf_amd import distributed as D
    from sudo_rm_rf_amd import optim
    dev = torch.device("cuda:0")
    variant, kw, T, fs, batch = bench.WORKLOADS[a.workload]
    torch.manual_seed(0)
    model = (imp.SuDORMRF if variant == "improved" else gc.GroupCommSudoRmRf)(**kw).to(dev).train()
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    opt = optim.FusedClipAdam(model.parameters(), lr=1e-3, clip_grad_norm=5.0)
    g = torch.Generator(device="cpu").manual_seed(1000)
    clean = torch.randn(batch, kw["num_sources"], T, generator=g).to(dev)
    mix = clean.sum(1, keepdim=True)
    mix = (mix - mix.mean(-1, keepdim=True)) / (mix.std(-1, keepdim=True) + 1e-9)

    def step(sync, acc):
        def mark(name, t0):
            if sync:
                torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            acc[name] = acc.get(name, 0.0) + (t1 - t0)
            return t1
        t = time.perf_counter()
        opt.zero_grad()
        t = mark("zero_grad", t)
        rec = model(mix)
        t = mark("forward", t)
        l = D.clamp_global_mean(loss_fn(rec, clean), min=-30., max=+30.)
        t = mark("loss", t)
        l.backward()
        t = mark("backward", t)
        D.allreduce_gradients(model.parameters())
        t = mark("allreduce", t)
        opt.step()
        t = mark("optimizer", t)

    for _ in range(3):
        step(False, {})
    torch.cuda.synchronize(dev)
    for sync in (False, True):
        acc = {}
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(sync, acc)
        issued = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        total = time.perf_counter() - t0
        print("%s: %.2f ms per step (host issue %.2f ms)" % ("synchronised after every section" if sync else "free-running", 1e3 * total / a.steps, 1e3 * issued / a.steps))
        for k, v in acc.items():
            print("   %-10s %7.2f ms" % (k, 1e3 * v / a.steps))


if __name__ == "__main__":
    main()
