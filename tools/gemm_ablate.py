#!/usr/bin/env python3
"""Ablation of the 8-wave split-bf16 GEMM pipeline on the proj_1x1 shape (diagnostics)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops
DEV = "cuda:0"
Bt, Cin, Cout, L = 32, 256, 512, 3200
g = torch.Generator().manual_seed(0)
x = torch.randn(Bt, Cin, L, generator=g).to(DEV)
w = (torch.randn(Cout, Cin, 1, generator=g) * Cin ** -0.5).to(DEV)
bias = torch.randn(Cout, generator=g).to(DEV)
import sudo_rm_rf_amd.ops as _o
_o.pack_pw_weight = lambda w_: None
names = {0: "full", 1: "no A loads", 2: "no B loads", 3: "no loads", 4: "no MFMA", 8: "no convert/store",
         12: "no MFMA, no convert", 15: "barriers + epilogue only"}
for abl, nm in names.items():
    ops.set_debug_flags(abl << 16)
    ops.pw_conv(x, w, bias); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.pw_conv(x, w, bias)
    e1.record(); torch.cuda.synchronize()
    print(f"ABL {abl:2d} {nm:28s} {e0.elapsed_time(e1) * 50:8.1f} us", flush=True)
ops.set_debug_flags(0)
