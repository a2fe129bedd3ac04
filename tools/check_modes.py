#!/usr/bin/env python3
"""bench.py's model / input for a workload: per-example max |difference| of the single-stream forward and of the
auto-tuned split forward against the generic kernels (kernel mode 1).  usage: check_modes.py [workload] [flags]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as gc  # noqa: E402
import sudo_rm_rf.dnn.models.improved_sudormrf as imp  # noqa: E402
from sudo_rm_rf_amd import ops  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_groupcomm_u8"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
variant, kw, T, fs, batch = bench.WORKLOADS[name]
torch.manual_seed(0)
model = (imp.SuDORMRF if variant == "improved" else gc.GroupCommSudoRmRf)(**kw).cuda().eval()
g = torch.Generator(device="cpu").manual_seed(1000)
wav = torch.randn(batch, 1, T, generator=g)
wav = ((wav - wav.mean(-1, keepdim=True)) / (wav.std(-1, keepdim=True) + 1e-9)).cuda()
eng = model._engine()
ops.set_debug_flags(flags)
with torch.no_grad():
    if os.environ.get("SEQ") == "bench":      # bench.py's order: tune, many back-to-back split forwards, then single
        split = model(wav)
        for _ in range(8):
            split = model(wav)
        eng.multi_stream = False
        single = model(wav)
        eng.multi_stream = True
    else:
        eng.multi_stream = False
        single = model(wav).clone()
        eng.multi_stream = True
        split = model(wav).clone()
    ops.set_kernel_mode(1)
    generic = model(wav).clone()
    ops.set_kernel_mode(0)
es = (single - generic).abs().amax(dim=(1, 2)).tolist()
ep = (split - generic).abs().amax(dim=(1, 2)).tolist()
print("split chosen:", eng._split_choice)
print("single vs generic: max %.3e, examples > 1e-4: %s" % (max(es), [i for i, e in enumerate(es) if e > 1e-4]))
print("split  vs generic: max %.3e, examples > 1e-4: %s" % (max(ep), [i for i, e in enumerate(ep) if e > 1e-4]))
