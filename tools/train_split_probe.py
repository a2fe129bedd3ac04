#!/usr/bin/env python3
"""Would the TRAINING step gain from the two-stream treatment the inference forward gets?  Times srf_forward_train + srf_backward
(random upstream gradient; no loss / optimizer) of one batch-32 call against two batch-16 calls running concurrently on two
streams (their flat gradients summed afterwards), cfg 2 (and cfg 4), through the C ABI."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf  # noqa: E402
from sudo_rm_rf_amd import _lib  # noqa: E402

DEV = torch.device("cuda", 0)
KW = {"cfg2": dict(out_channels=256, in_channels=512, num_blocks=16, upsampling_depth=5, enc_kernel_size=21, enc_num_basis=512,
                   num_sources=2),
      "cfg4": dict(out_channels=512, in_channels=512, num_blocks=36, upsampling_depth=6, enc_kernel_size=21, enc_num_basis=2048,
                   num_sources=2)}


def main():
    lib = _lib.load()
    T = 32000
    for name in sys.argv[1:] or ["cfg2"]:
        model = improved_sudormrf.SuDORMRF(**KW[name]).to(DEV).train()
        eng = model._engine()
        params = [p.detach() for p in model.state_dict(keep_vars=True).values()]
        ptab = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        total = sum(p.numel() for p in params)
        g = torch.Generator(device=DEV).manual_seed(0)
        x = torch.randn(32, 1, T, generator=g, device=DEV)
        gout = torch.randn(32, 2, T, generator=g, device=DEV) * 1e-3

        class Part:
            def __init__(self, lo, hi, lane):
                self.lo, self.hi = lo, hi
                self.plan = eng.plan_for(hi - lo, T, DEV, lane=lane)
                sb, cb = self.plan.train_sizes()
                self.saved = torch.empty(sb, dtype=torch.uint8, device=DEV)
                self.scratch = torch.empty(cb, dtype=torch.uint8, device=DEV)
                self.out = torch.empty(hi - lo, 2, T, device=DEV)
                self.flat = torch.zeros(total, device=DEV)
                off, self.gtab = 0, None
                ptrs = []
                for p in params:
                    ptrs.append(self.flat.data_ptr() + 4 * off)
                    off += p.numel()
                self.gtab = (C.c_void_p * len(params))(*ptrs)
                self.x = x[lo:hi].contiguous()
                self.g = gout[lo:hi].contiguous()

            def run(self):
                st = _lib.current_stream(DEV)
                self.flat.zero_()
                _lib.check(lib.srf_forward_train(self.plan.handle, ptab, len(params), _lib.ptr(self.x), _lib.ptr(self.out),
                                                 _lib.ptr(self.saved), self.saved.numel(), _lib.ptr(self.scratch),
                                                 self.scratch.numel(), st), "fwd")
                _lib.check(lib.srf_backward(self.plan.handle, ptab, self.gtab, len(params), _lib.ptr(self.x), _lib.ptr(self.g),
                                            _lib.ptr(self.saved), self.saved.numel(), _lib.ptr(self.scratch),
                                            self.scratch.numel(), st), "bwd")

        whole = Part(0, 32, 11)
        halves = [Part(0, 16, 12), Part(16, 32, 13)]
        skew = [Part(0, 20, 14), Part(20, 32, 15)]
        streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]

        def step_whole():
            whole.run()

        def step_split(parts):
            cur = torch.cuda.current_stream(DEV)
            for s, p in zip(streams, parts):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    p.run()
            for s in streams:
                cur.wait_stream(s)
            parts[0].flat.add_(parts[1].flat)

        def timeit(fn, n=8):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        for rep in range(2):
            a = timeit(step_whole)
            b = timeit(lambda: step_split(halves))
            c = timeit(lambda: step_split(skew))
            print("%s rep %d: whole batch %.2f ms | two halves on two streams %.2f ms (%+.1f %%) | 20 + 12 %.2f ms (%+.1f %%)"
                  % (name, rep, a, b, 100 * (b / a - 1), c, 100 * (c / a - 1)), flush=True)
        # the summed gradient equals the whole-batch gradient up to rounding (the upstream gradient is per example)
        step_whole()
        step_split(halves)
        torch.cuda.synchronize()
        num = float((whole.flat - halves[0].flat).norm())
        print("   |g_whole - (g_a + g_b)| / |g_whole| = %.2e" % (num / float(whole.flat.norm())))
        del model, whole, halves, skew


if __name__ == "__main__":
    main()
