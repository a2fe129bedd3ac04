#!/usr/bin/env python3
"""Experiment: one batch-32 forward vs N concurrent sub-batch forwards on N streams (examples are independent)."""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf
import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as gcm
from oracle.schema import CONFIGS
DEV = "cuda:0"
from sudo_rm_rf_amd import ops
ops.set_debug_flags(int(os.environ.get("SRF_FLAGS", "0")))
os.environ["SRF_STREAM_SPLIT"] = "off"
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_improved_u16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = CONFIGS[name]
torch.manual_seed(0)
cls = improved_sudormrf.SuDORMRF if cfg.variant == "improved" else gcm.GroupCommSudoRmRf
model = cls(**cfg.ctor_kwargs()).to(DEV).eval()
T = 128000 if "n4096" in name else 32000
wav = torch.randn(B, 1, T, device=DEV)
models = [model] + [copy.deepcopy(model) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(4)]
def run(splits):
    cur = torch.cuda.current_stream()
    outs = []
    lo = 0
    for i, n in enumerate(splits):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs.append(models[i](wav[lo:lo + n]))
        lo += n
    for i in range(len(splits)):
        cur.wait_stream(streams[i])
    return outs
with torch.no_grad():
    cases = [[B], [B // 2, B // 2], [B // 4] * 4, [B // 2 + B // 8, B // 2 - B // 8]]
    if B % 3 == 0: cases.append([B // 3] * 3)
    for sp in cases:
        fn = (lambda: model(wav)) if len(sp) == 1 else (lambda: run(sp))
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        print("%-10s %-18s %.3f ms" % (name[:10], sp, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
