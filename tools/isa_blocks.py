#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a gfx950 device assembly file: where the MFMAs, the vector-memory and
LDS instructions and -- above all -- the scratch (spill) traffic sit.  A spill in a set-up block costs nothing; one inside a
k-loop block costs a vector-memory slot per iteration.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -S --cuda-device-only csrc/X.hip -o /tmp/X.s
    python tools/isa_blocks.py /tmp/X.s <substring of the mangled kernel name> [min instructions per block]"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path, errors="replace").read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[_A-Za-z]\S*:", l) and key in l and not l.startswith(".L"):
            start = i
            break
    if start is None:
        raise SystemExit("kernel containing %r not found" % key)
    end = next((i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
    print(lines[start].split(":")[0])
    blocks, cur = [], ("entry", [])
    for l in lines[start + 1:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = (l.split(":")[0], [])
        elif re.match(r"\s+[a-z]", l):
            cur[1].append(l.strip())
    blocks.append(cur)
    tot = {}
    print("%-12s %6s %6s %6s %5s %5s %5s %5s %5s" % ("block", "instr", "mfma", "valu", "ds", "vmem", "salu", "sst", "sld"))
    for lab, ls in blocks:
        c = {
            "instr": len(ls),
            "mfma": sum(x.startswith("v_mfma") for x in ls),
            "valu": sum(x.startswith("v_") and not x.startswith("v_mfma") for x in ls),
            "ds": sum(x.startswith("ds_") for x in ls),
            "vmem": sum(bool(re.match(r"(buffer_|global_|flat_)", x)) for x in ls),
            "salu": sum(x.startswith("s_") for x in ls),
            "sst": sum(x.startswith("scratch_store") for x in ls),
            "sld": sum(x.startswith("scratch_load") for x in ls),
        }
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        if c["instr"] >= min_n or c["sst"] or c["sld"]:
            print("%-12s %6d %6d %6d %5d %5d %5d %5d %5d" % (lab, c["instr"], c["mfma"], c["valu"], c["ds"], c["vmem"], c["salu"], c["sst"], c["sld"]))
    print("%-12s %6d %6d %6d %5d %5d %5d %5d %5d" % ("total", tot["instr"], tot["mfma"], tot["valu"], tot["ds"], tot["vmem"], tot["salu"], tot["sst"], tot["sld"]))


if __name__ == "__main__":
    main()
