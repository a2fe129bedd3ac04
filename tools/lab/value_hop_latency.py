#!/usr/bin/env python3
"""Latency of a cross-stream dependency made of stream MEMORY operations (hipStreamWriteValue64 / hipStreamWaitValue64 on a
hipMallocSignalMemory word, BETA API) against HIP events (tools/lab/event_hop_latency.py: ~14 us per hop)."""
import ctypes as C
import torch
dev = "cuda:0"
hip = C.CDLL("libamdhip64.so")
x = torch.zeros(64, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
attr = C.c_int(0)
# hipDeviceAttributeCanUseStreamWaitValue: look the enum value up by probing is fragile; just try the calls
sig = C.c_void_p()
rc = hip.hipExtMallocWithFlags(C.byref(sig), C.c_size_t(8), C.c_uint(2))
print("hipExtMallocWithFlags(signal) rc", rc, hex(sig.value or 0), flush=True)
assert rc == 0
hip.hipStreamWriteValue64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint]
hip.hipStreamWaitValue64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_uint64]
N = 400
base = [0]


def pingpong():
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.randn(8192, 8192, device=dev)
    v0 = base[0]
    with torch.cuda.stream(sa):
        for _ in range(6):
            big @ big
        e0.record(sa)
        assert hip.hipStreamWriteValue64(C.c_void_p(sa.cuda_stream), sig, v0 + 1, 0) == 0
    for k in range(N):
        st = sa if k % 2 == 0 else sb
        with torch.cuda.stream(st):
            assert hip.hipStreamWaitValue64(C.c_void_p(st.cuda_stream), sig, v0 + 1 + k, 0, 0xFFFFFFFFFFFFFFFF) == 0   # Gte
            x.add_(1.0)
            assert hip.hipStreamWriteValue64(C.c_void_p(st.cuda_stream), sig, v0 + 2 + k, 0) == 0
    last = sa if (N - 1) % 2 == 0 else sb
    sa.wait_stream(last)
    e1.record(sa)
    torch.cuda.synchronize()
    base[0] = v0 + N + 2
    return e0.elapsed_time(e1) * 1e3 / N


for _ in range(3):
    print("ping-pong over two streams with stream value operations %.2f us per hop" % pingpong(), flush=True)
