#!/usr/bin/env python3
"""Bisect the GroupComm two-stream corruption (DESIGN.md open issue 1).

For every configuration (debug flags / kernel mode) run N back-to-back forwards of bench.py's cfg-3 model with the
batch split over two streams, compare every forward with the single-stream forward of the same configuration, and
-- for the stage attribution -- compare the intermediates each lane's plan keeps in its workspace (encoder output,
last block output, masked encoding) with the single-stream plan's.
usage: diag_gc_split.py [workload] [N] [split, e.g. 20:12]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import sudo_rm_rf.dnn.models.groupcomm_sudormrf_v2 as gc  # noqa: E402
import sudo_rm_rf.dnn.models.improved_sudormrf as imp  # noqa: E402
from sudo_rm_rf_amd import engine as eng_mod, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_groupcomm_u8"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
split = sys.argv[3] if len(sys.argv) > 3 else "5:3"
variant, kw, T, fs, batch = bench.WORKLOADS[name]
torch.manual_seed(0)
model = (imp.SuDORMRF if variant == "improved" else gc.GroupCommSudoRmRf)(**kw).cuda().eval()
g = torch.Generator(device="cpu").manual_seed(1000)
wav = torch.randn(batch, 1, T, generator=g)
wav = ((wav - wav.mean(-1, keepdim=True)) / (wav.std(-1, keepdim=True) + 1e-9)).cuda()
eng = model._engine()
dev = wav.device
L = None


def fetch(plan, lo, n):
    c = eng.cfg_tuple
    N_, B_ = kw["enc_num_basis"], kw["out_channels"]
    Lp = plan.frames
    return [plan.debug_fetch(0, (n, N_, Lp)), plan.debug_fetch(1, (n, B_, Lp))]


CONFIGS = [
    ("default", 0, 0),
    ("tac_one_step(1024)", 1024, 0),
    ("tac_generic(1<<24)", 1 << 24, 0),
    ("pyramid_unfused(16)", 16, 0),
    ("pyramid_pass1_nonpersistent(128)", 128, 0),
    ("pyramid_lds(64)", 64, 0),
    ("gemm_pointer(1<<27)", 1 << 27, 0),
    ("gemm_one_tile(2048)", 2048, 0),
    ("generic_kernels(mode1)", 0, 1),
]
only = os.environ.get("ONLY")
report = {}
with torch.no_grad():
    for cname, flags, mode in CONFIGS:
        if only and only not in cname:
            continue
        ops.set_debug_flags(flags)
        ops.set_kernel_mode(mode)
        eng._plans.clear()
        eng._split_choice.clear()
        eng.multi_stream = False
        ref = model(wav).clone()
        rplan = eng.last_plan
        rint = fetch(rplan, 0, batch)
        ref2 = model(wav)
        rerun = (ref2 - ref).abs().max().item()
        eng.multi_stream = True
        eng_mod._SPLIT_MODE = split
        bad_events = []
        stage_hits = {"enc": 0, "blocks": 0, "tail_only": 0}
        for it in range(N):
            out = model(wav)
            err = (out - ref).abs().amax(dim=(1, 2))
            bad = (err > 2e-5).nonzero().flatten().tolist()
            if bad:
                # stage attribution from the lanes' workspaces (still hold this forward's intermediates)
                parts = eng._split_candidates(batch)[0]
                lo = 0
                st = []
                for lane, n in enumerate(parts):
                    plan = eng.plan_for(n, T, dev, lane=lane + 1)
                    ints = fetch(plan, lo, n)
                    e_enc = (ints[0] - rint[0][lo:lo + n]).abs().amax(dim=(1, 2))
                    e_blk = (ints[1] - rint[1][lo:lo + n]).abs().amax(dim=(1, 2))
                    for i in range(n):
                        if lo + i in bad:
                            st.append((lo + i, lane, float(e_enc[i]), float(e_blk[i]), float(err[lo + i])))
                            if e_enc[i] > 1e-6:
                                stage_hits["enc"] += 1
                            elif e_blk[i] > 2e-6:
                                stage_hits["blocks"] += 1
                            else:
                                stage_hits["tail_only"] += 1
                    lo += n
                bad_events.append((it, st))
        eng_mod._SPLIT_MODE = "auto"
        nb = sum(len(s) for _, s in bad_events)
        print("%-36s single-stream rerun diff %.1e | split %s: %d bad forwards of %d, %d bad examples, stages %s" %
              (cname, rerun, split, len(bad_events), N, nb, stage_hits), flush=True)
        for it, st in bad_events[:6]:
            print("     it %d: (example, lane, enc err, last-block err, out err) %s" %
                  (it, [(a, b, "%.1e" % c, "%.1e" % d, "%.1e" % e) for a, b, c, d, e in st[:8]]), flush=True)
        report[cname] = {"bad_forwards": len(bad_events), "bad_examples": nb, "stages": stage_hits, "N": N}
# ---- the same lanes (plans, workspaces, sub-batch sizes) launched one after the other on ONE stream: concurrency or shape?
if not only:
    with torch.no_grad():
        ops.set_debug_flags(0)
        ops.set_kernel_mode(0)
        eng._plans.clear()
        eng.multi_stream = False
        ref = model(wav).clone()
        weights = [p.detach() for p in eng_mod._weights(model)]
        table = eng._param_table(weights, dev)
        for parts in ((20, 12), (12, 20), (16, 16)):
            out = torch.empty_like(ref)
            nbad = 0
            for it in range(N):
                lo = 0
                for lane, n in enumerate(parts):
                    eng.plan_for(n, T, dev, lane=lane + 1).forward(table, wav[lo:lo + n], out[lo:lo + n])
                    lo += n
                err = (out - ref).abs().amax(dim=(1, 2))
                nbad += int((err > 2e-5).sum())
            print("serial lanes %s on one stream: %d bad examples in %d forwards" % (parts, nbad, N), flush=True)
            report["serial_%d_%d" % parts] = nbad
        # concurrent, other splits
        for sp in ("1:1", "3:5"):
            eng._split_choice.clear()
            eng.multi_stream = True
            eng_mod._SPLIT_MODE = sp
            nbad = 0
            for it in range(N):
                err = (model(wav) - ref).abs().amax(dim=(1, 2))
                nbad += int((err > 2e-5).sum())
            print("concurrent split %s: %d bad examples in %d forwards" % (sp, nbad, N), flush=True)
            report["concurrent_" + sp] = nbad
        eng_mod._SPLIT_MODE = "auto"
ops.set_debug_flags(0)
ops.set_kernel_mode(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(report, open(os.path.join(ROOT, "gpurun_out", "diag_gc_split_%s.json" % name), "w"), indent=1)
