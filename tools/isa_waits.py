#!/usr/bin/env python3
"""Compile one HIP source for gfx950 with -save-temps and list, per kernel, every `s_waitcnt vmcnt(N)` together with the
number of VMEM loads the compiler can see in flight at that point (loads issued since the previous full drain, same basic
block chain, textual order) -- a quick way to spot the waits that drain a software pipeline early: a `vmcnt(0)` a few
instructions after a batch of buffer loads is a full memory round trip on every wavefront.  (Round 3: this is how the
per-tile `vmcnt(0)` behind the GlobLN statistics load and the every-second-step `vmcnt(0)` of the PRO 0 GEMM were found.)

    python tools/isa_waits.py sudo_rm_rf_amd/csrc/srf_pwconv_x3v.hip [kernel-name-substring] [max N]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sudo_rm_rf_amd import build as B  # noqa: E402


def main():
    src = os.path.abspath(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    base = os.path.basename(src)
    with tempfile.TemporaryDirectory() as td:
        cmd = [B.hipcc()] + B.FLAGS + B.FILE_FLAGS.get(base, []) + ["-save-temps=obj", "-c", src, "-o", os.path.join(td, "x.o")]
        subprocess.run(cmd, check=True, capture_output=True)
        asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
        lines = open(os.path.join(td, asm)).read().splitlines()
        keep = os.environ.get("ISA_KEEP")
        if keep:
            open(keep, "w").write("\n".join(lines))
    name, since, n_mfma = None, [], 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, since = m.group(1), []
            show = pat in name
            if show:
                print("==", name)
            continue
        if name is None or not show:
            continue
        t = l.strip()
        if re.match(r"(buffer_load|global_load|flat_load)", t):
            since.append((i, t.split()[0]))
        m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
        if m and int(m.group(1)) <= nmax:
            recent = [k for k, _ in since if i - k < 400]
            print("  line %6d  %-34s  %2d loads in the 400 lines before (nearest %s lines back)"
                  % (i + 1, t, len(recent), (i - recent[-1]) if recent else "-"))


if __name__ == "__main__":
    main()
