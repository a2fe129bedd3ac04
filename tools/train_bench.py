#!/usr/bin/env python3
"""Time one training step (zero_grad, forward, PIT-SI-SDR, clamp, backward, clip_grad_norm_, Adam.step -- the body
of the reference runner's loop, run_improved_sudormrf.py:146-177) on one MI355X, with the in-library profiler's
per-kernel breakdown.  usage: train_bench.py [cfg2_improved_u16|cfg4_improved_u36_n2048] [batch] [steps]"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sudo_rm_rf.dnn.losses.sisdr as sisdr_lib  # noqa: E402
import sudo_rm_rf.dnn.experiments.utils.mixture_consistency as mixture_consistency  # noqa: E402
import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as sudormrf_gc_v2  # noqa: E402
import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf  # noqa: E402
from oracle.schema import CONFIGS  # noqa: E402
from sudo_rm_rf_amd import _lib  # noqa: E402

DEV = "cuda:0"


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_improved_u16"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cfg = CONFIGS[name]
    T = 32000
    torch.manual_seed(0)
    gc = cfg.variant == "groupcomm"
    model = (sudormrf_gc_v2.GroupCommSudoRmRf if gc else improved_sudormrf.SuDORMRF)(**cfg.ctor_kwargs()).to(DEV)
    loss_fn = sisdr_lib.PITLossWrapper(sisdr_lib.PairwiseNegSDR("sisdr"), pit_from='pw_mtx')
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1)
    clean = torch.randn(batch, cfg.num_sources, T, generator=g).to(DEV)
    mix = clean.sum(1, keepdim=True)
    mix = (mix - mix.mean(-1, keepdim=True)) / (mix.std(-1, keepdim=True) + 1e-9)
    # targets correlated with the estimate so that the loss is not pinned at the +30 clamp
    model.train()

    def step():
        opt.zero_grad()
        rec = model(mix)
        if gc:
            rec = mixture_consistency.apply(rec, mix)
        l = torch.clamp(loss_fn(rec, clean), min=-30., max=+30.)
        l.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        return l

    for _ in range(2):
        l = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            model(mix)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            model(mix)
        torch.cuda.synchronize()
        dinf = (time.perf_counter() - t1) / steps
    # per-kernel breakdown of one training step
    model.train()
    lib = _lib.load()
    stream = _lib.current_stream(torch.device(DEV))
    lib.srf_profile_begin(stream)
    step()
    cnt = C.c_int(0)
    _lib.check(lib.srf_profile_end(stream, C.byref(cnt)), "srf_profile_end")
    per = {}
    seq = {}
    nm, ms = C.c_char_p(), C.c_float()
    for i in range(cnt.value):
        lib.srf_profile_get(i, C.byref(nm), C.byref(ms))
        if nm.value.startswith(b"("):        # "(gap)": host-side idle, not a kernel
            continue
        e = per.setdefault(nm.value.decode(), [0.0, 0])
        e[0] += ms.value
        e[1] += 1
        seq.setdefault(nm.value.decode(), []).append(ms.value * 1e3)
    # TRAIN_SEQ="family:period,...": the family's launches in order, averaged by position within a period (e.g. the five
    # depthwise-backward launches of a U-ConvBlock: "dwconv5_bwd:5") -- to stderr
    for item in filter(None, os.environ.get("TRAIN_SEQ", "").split(",")):
        fam, period = item.split(":")
        v, period = seq.get(fam, []), int(period)
        pos = [[x for j, x in enumerate(v) if j % period == k] for k in range(period)]
        print("%s (%d launches) us by position: %s" % (fam, len(v), "  ".join("%.1f" % (sum(q) / max(len(q), 1)) for q in pos)),
              file=sys.stderr)
    plan = model._engine().last_plan
    saved, scratch = plan.train_sizes()
    out = {"workload": "%s training step, batch %d, T=%d" % (name, batch, T), "ms_per_step": dt * 1e3,
           "trained_seconds_per_sec": batch * (T / 8000.0) / dt, "inference_ms": dinf * 1e3,
           "train_over_inference": dt / dinf, "loss": float(l), "saved_GB": saved / 2 ** 30, "scratch_GB": scratch / 2 ** 30,
           "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
           "kernels_ms": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in
                          sorted(per.items(), key=lambda kv: -kv[1][0])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
