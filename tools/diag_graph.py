#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import sudo_rm_rf.dnn.models.improved_sudormrf as imp
from sudo_rm_rf_amd import engine as engine_mod
variant, kw, T, fs, batch = bench.WORKLOADS["cfg1_improved_u8"]
torch.manual_seed(0)
model = imp.SuDORMRF(**kw).cuda().eval()
wav = torch.randn(1, 1, T, device="cuda")
with torch.no_grad():
    engine_mod._GRAPH_MODE = "off"
    eager = model(wav).clone()
    engine_mod._GRAPH_MODE = "auto"
    for i in range(12):
        o = model(wav)
        e = float((o - eager).abs().max())
        print(i, "graphs", len(model._engine()._graphs), "err %.3e" % e, flush=True)
    outs = [model(wav) for _ in range(50)]
    torch.cuda.synchronize()
    print("50 back-to-back: worst err %.3e" % max(float((o - eager).abs().max()) for o in outs))
    fin = bool(torch.isfinite(outs[-1]).all())
    o = model(wav)
    print("after isfinite: err %.3e" % float((o - eager).abs().max()))
    big = torch.randn(64, 1024, 1024, device="cuda"); del big
    o = model(wav)
    print("after alloc/free: err %.3e" % float((o - eager).abs().max()))
    from sudo_rm_rf_amd import ops
    ops.set_kernel_mode(1); g1 = model(wav[:1]); ops.set_kernel_mode(0)
    print("mode-1 call (replays the mode-0 graph): err %.3e" % float((g1 - eager).abs().max()))
    o = model(wav)
    print("after: err %.3e" % float((o - eager).abs().max()))
