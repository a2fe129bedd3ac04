#!/usr/bin/env python3
"""Infinity-Cache (256 MiB L3) probe: does a consumer kernel run faster when its input was produced
immediately before and is small enough to still sit in the L3?  FETCH_SIZE cannot tell (L3 hits are counted), so this is timing.

Part 1 (raw): producer = srf_gln_apply (reads X, writes Y of the same size), consumer = srf_gln_stats(Y) (a pure streaming
read of this library); the consumer is timed with events, (a) right behind its producer, (b) behind a 1-GiB flush of unrelated
data.  Sizes 13 ... 420 MB.
Part 2 (the forward's own kernels, single stream, in-library profiler): per-example time of every kernel family of cfg 2 at
batch 4 / 8 / 12 / 16 / 24 / 32 -- the producer of every kernel's input is the launch in front of it.

    python tools/l3_probe.py [--train]      -> text table on stdout (profiles/r06_l3_probe.txt)"""
import json
import os
import statistics
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"


def part1():
    print("== part 1: consumer (srf_gln_stats, streaming read) right behind its producer vs behind a 1-GiB flush")
    print("%8s %12s %12s %10s %10s" % ("MB", "hot us", "cold us", "hot TB/s", "cold TB/s"))
    flush_src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=DEV).normal_()
    C, L = 512, 3200
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    for Bt in (2, 4, 8, 12, 16, 24, 32, 64):
        x = torch.randn(Bt, C, L, device=DEV)
        sums = ops.gln_stats(x, Bt)
        mb = x.numel() * 4 / 1e6
        res = {}
        for mode in ("hot", "cold"):
            ts = []
            for _ in range(12):
                y = ops.gln_apply(x, sums, gamma, beta)          # producer: writes y (mb MB)
                if mode == "cold":
                    flush_src.add_(1.0)                          # 2 GiB of unrelated traffic
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gln_stats(y, Bt)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
                del y
            res[mode] = statistics.median(ts[2:])
        print("%8.1f %12.1f %12.1f %10.2f %10.2f" % (mb, res["hot"], res["cold"], mb / res["hot"], mb / res["cold"]), flush=True)


def part2(train):
    print("== part 2: cfg 2 %s, single stream, per-kernel us PER EXAMPLE by batch (in-library profiler)" %
          ("training step" if train else "forward"))
    rows = {}
    batches = (4, 8, 12, 16, 24, 32)
    for b in batches:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(b), "--steps", "10", "--warmup", "4",
               "--no-cpu-baseline"] + (["--train"] if train else [])
        env = dict(os.environ, SRF_STREAM_SPLIT="off")
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            print("batch", b, "failed:", e, out.stderr[-400:])
            continue
        rows[("ms_per_step", "")] = rows.get(("ms_per_step", ""), {})
        rows[("ms_per_step", "")][b] = d["ms_per_step"] * 1e3 / b
        for name, k in d.get("kernels", {}).items():
            n = k.get("launches_per_forward", k.get("launches_per_step", 1))
            rows.setdefault((name, n), {})[b] = k["avg_launch_us"] / b
    print("%-34s %5s " % ("kernel (us per example per launch)", "n") + " ".join("%8s" % ("bs%d" % b) for b in batches) + "   bs8/bs32")
    for (name, n), v in sorted(rows.items(), key=lambda kv: -kv[1].get(32, 0) * (kv[0][1] or 1)):
        ratio = v[8] / v[32] if 8 in v and 32 in v and v[32] else float("nan")
        print("%-34s %5s " % (name, n) + " ".join("%8.2f" % v[b] if b in v else "%8s" % "-" for b in batches) + "   %.2f" % ratio)


if __name__ == "__main__":
    part1()
    part2("--train" in sys.argv)
