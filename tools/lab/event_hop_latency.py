#!/usr/bin/env python3
"""Latency of a cross-stream event dependency on this box: N tiny kernels ping-ponged between two streams (each waits for an event
recorded after the previous one on the OTHER stream) against the same N kernels on one stream."""
import torch
dev = "cuda:0"
x = torch.zeros(64, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 400


def one():
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.randn(8192, 8192, device=dev)
    with torch.cuda.stream(sa):
        for _ in range(6):
            big @ big
        e0.record(sa)
        for _ in range(N):
            x.add_(1.0)
        e1.record(sa)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


def pingpong():
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.randn(8192, 8192, device=dev)
    with torch.cuda.stream(sa):
        for _ in range(6):
            big @ big          # ~ tens of ms: the host enqueues the whole chain behind it
        e0.record(sa)
        prev = torch.cuda.Event()
        prev.record(sa)
    for k in range(N):
        st = sa if k % 2 == 0 else sb
        with torch.cuda.stream(st):
            if prev is not None:
                st.wait_event(prev)
            x.add_(1.0)
            prev = torch.cuda.Event()
            prev.record(st)
    sa.wait_event(prev)
    e1.record(sa)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


for _ in range(3):
    print("same stream %.2f us per kernel; ping-pong over two streams with events %.2f us per kernel" % (one(), pingpong()), flush=True)
