#!/usr/bin/env python3
"""Copy the judged artefacts of one tools/gpu_profiles.sh run (gpurun_out/<dir>) into profiles/ (tracked), as <tag>_*:
GPU test summary, bench lines of the five configurations (+ the exact-fp32 line), training steps, rocprofv3 kernel stats,
HBM-traffic and SQ-counter summaries of the forward (cfgs 2 / 4 / 5) and of the training step (cfgs 2 / 4), the rocm-smi power
samples.  usage: tools/collect_profiles.py <gpurun_out dir name> [tag, default r05]"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", sys.argv[1])
T = sys.argv[2] if len(sys.argv) > 2 else "r05"
P = os.path.join(ROOT, "profiles")
SHORT = {"cfg1_improved_u8": "cfg1_bs1", "cfg2_improved_u16": "cfg2_bs32", "cfg3_groupcomm_u8": "cfg3_groupcomm_bs32",
         "cfg4_improved_u36_n2048": "cfg4_u36_n2048_bs32", "cfg5_improved_u36_n4096": "cfg5_u36_n4096_8s16k_bs16"}


def cp(src, dst):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(P, dst))
        print("  ", dst)


def kname(s):
    return s.split("(")[0].replace("void ", "")


for w, short in SHORT.items():
    cp(os.path.join(R, "bench_%s.json" % w), "%s_%s_bench.json" % (T, short))
    cp(os.path.join(R, "train_%s.json" % w), "%s_%s_train_step_bs32.json" % (T, w))
cp(os.path.join(R, "power.log"), "%s_power_rocm_smi_final.log" % T)
cp(os.path.join(R, "bench_cfg2_exact_fp32.json"), "%s_cfg2_bs32_bench_exact_fp32_kernel_mode2.json" % T)
cp(os.path.join(R, "env.log"), "%s_env.log" % T)
if os.path.exists(os.path.join(R, "pytest_gpu.log")):
    tail = open(os.path.join(R, "pytest_gpu.log")).read().splitlines()[-6:]
    open(os.path.join(P, "%s_pytest_gpu_summary.txt" % T), "w").write(
        "# python -m pytest tests -q -m gpu on the box the rest of this set was measured on (tools/gpu_profiles.sh)\n" + "\n".join(tail) + "\n")
    print("  ", "%s_pytest_gpu_summary.txt" % T)
JOBS = [(w, short, "--workload %s" % w) for w, short in SHORT.items()]
JOBS += [("train_" + w, SHORT[w] + "_train_step", "--train --workload %s" % w) for w in ("cfg2_improved_u16", "cfg3_groupcomm_u8", "cfg4_improved_u36_n2048")]
for w, short, cmdline in JOBS:
    stats = glob.glob(os.path.join(R, "prof_%s" % w, "**", "*kernel_stats.csv"), recursive=True)
    dur = {}
    if stats:
        rows = [l for l in open(stats[0]) if "at::native" not in l]
        open(os.path.join(P, "%s_%s_rocprofv3_kernel_stats.csv" % (T, short)), "w").writelines(rows)
        print("  ", "%s_%s_rocprofv3_kernel_stats.csv" % (T, short))
        for r in csv.DictReader(open(stats[0])):
            dur[kname(r["Name"])] = float(r["AverageNs"]) / 1e3
    # ---- HBM traffic (passes 1, 2) and SQ counters (passes 3, 4)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in sorted(glob.glob(os.path.join(R, "pmc_%s_*" % w, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "srf_" not in r["Kernel_Name"]:
                continue
            e = agg[kname(r["Kernel_Name"])][r["Counter_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    if not agg:
        continue
    with open(os.path.join(P, "%s_%s_pmc_hbm_traffic.csv" % (T, short)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of `bench.py %s --steps 2 "
                "--warmup 1`); counter unit KB per launch.\n# gfx950 (MI355X_MICROARCH.md): FETCH_SIZE reports 1/2 of the bytes of wide "
                "coalesced reads -> corrected = 2 * FETCH_SIZE; WRITE_SIZE as reported.\n" % cmdline)
        f.write("counter,kernel,launches,avg_KB_per_launch,corrected_MB_per_launch\n")
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            for k, cs in sorted(agg.items()):
                if c in cs:
                    n, v = cs[c]
                    f.write('%s,"%s",%d,%.1f,%.1f\n' % (c, k, n, v / n, (2 * v / n if c == "FETCH_SIZE" else v / n) / 1024))
    print("  ", "%s_%s_pmc_hbm_traffic.csv" % (T, short))
    with open(os.path.join(P, "%s_%s_pmc_sq_mfma_valu_lds.txt" % (T, short)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc <group> passes of `bench.py %s --steps 2 --warmup 1` (single stream), averages per launch.\n"
                "# SIMDs = 1024 (256 CUs x 4).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES is summed over\n"
                "# 32 shader engines: kernel cycles ~ SQ_BUSY_CYCLES / 32, MFMA utilisation = MFMA_BUSY / (1024 x that).\n" % cmdline)
        order = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [1, 0])[1])
        for k, cs in order:
            v = {c: x[1] / x[0] for c, x in cs.items() if c not in ("FETCH_SIZE", "WRITE_SIZE")}
            if not v:
                continue
            n = max(x[0] for x in cs.values())
            line = "%-70s n=%-4d" % (k[:70], n)
            if k in dur:
                line += " %8.1f us" % dur[k]
            f.write(line + "\n    " + "  ".join("%s=%.4g" % (c, v[c]) for c in sorted(v)) + "\n")
            if v.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                cyc = v["SQ_BUSY_CYCLES"] / 32
                mhz = (cyc / dur[k]) if k in dur else 0.0
                f.write("    -> kernel ~%.0f k cycles%s; MFMA pipe busy %.1f %% of SIMD-cycles; VALU instr / SIMD = %.1f k; bf16 MFMA ops = %.3g\n"
                        % (cyc / 1e3, (" = %.0f MHz at the traced duration" % mhz) if mhz else "",
                           100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), v.get("SQ_INSTS_VALU", 0) / 1024e3,
                           v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0)))
    print("  ", "%s_%s_pmc_sq_mfma_valu_lds.txt" % (T, short))
# ---- round-5 extras (part x of tools/gpu_profiles.sh)
cp(os.path.join(R, "pair_ab.log"), "%s_pair_ab_final_build.log" % T)
cp(os.path.join(R, "two_stream_events.txt"), "%s_cfg2_two_stream_timeline_final_build.txt" % T)
cp(os.path.join(R, "two_stream_events.json"), "%s_cfg2_two_stream_timeline_final_build.json" % T)
cp(os.path.join(R, "pair_power_probe.log"), "%s_pair_zero_vs_random_operands.log" % T)
