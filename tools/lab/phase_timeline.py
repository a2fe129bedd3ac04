#!/usr/bin/env python3
"""Raw per-stream mark timeline of ONE steady-state two-stream forward (phase lock on / off): name, end time, stream."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa
import two_stream_events as T  # noqa
from sudo_rm_rf_amd import _lib  # noqa
import sudo_rm_rf.dnn.models.improved_sudormrf as imp  # noqa

dev = torch.device("cuda:0")
lib = _lib.load()
variant, kw, Tn, fs, batch = bench.WORKLOADS["cfg2_improved_u16"]
torch.manual_seed(0)
model = imp.SuDORMRF(**kw).to(dev).eval()
wav = torch.randn(batch, 1, Tn, device=dev)
with torch.no_grad():
    for _ in range(12):
        model(wav)
marks, ms = T.collect(lib, model, wav, 6, dev)
print("ms per forward (instrumented)", ms, "split", model._engine()._split_choice, model._engine()._split_lock)
# the 4th forward: between the 4th and 5th "(gap)" of stream 0
gaps = [i for i, m in enumerate(marks) if m[0].startswith("(") ]
byt = sorted(marks, key=lambda m: m[1])
g = sorted(m[1] for m in marks if m[0].startswith("("))
lo, hi = g[6], g[8] if len(g) > 8 else 1e9
last = {}
for name, t, st in byt:
    if lo <= t < hi:
        print("%9.1f us  stream %d  %-28s (+%6.1f)" % ((t - lo) * 1e3, st, name, (t - last.get(st, lo)) * 1e3))
        last[st] = t
