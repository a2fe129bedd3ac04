#!/usr/bin/env python3
"""Batch-32 cfg 2 forward against the golden outputs for a list of debug-flag values (single stream unless
SRF_STREAM_SPLIT says otherwise): localises which kernel family a parity regression comes from.
usage: check_bs32.py [flags ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import json  # noqa: E402
from conftest import load_case  # noqa: E402
import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf  # noqa: E402
from sudo_rm_rf_amd import ops  # noqa: E402

man = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST.json")))
cfg, sd, wav, gold = load_case(man, "cfg2_improved_u16")
m = improved_sudormrf.SuDORMRF(**cfg.ctor_kwargs())
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda().eval()
for batch in (32, 16):
    x = torch.from_numpy(np.concatenate([wav] * (batch // 2), 0)).cuda()
    for f in [int(a) for a in sys.argv[1:]] or [0]:
        ops.set_debug_flags(f)
        with torch.no_grad():
            out = m(x).cpu().numpy()
        err = [float(np.abs(out[i] - gold["out"][i % 2]).max()) for i in range(batch)]
        bad = [i for i, e in enumerate(err) if e > 1e-4]
        print("batch %d flags %d: max err %.3e, examples over 1e-4: %s" % (batch, f, max(err), bad))
ops.set_debug_flags(0)
