"""Build libsudormrf_hip.so (gfx950) in-tree with hipcc.  No JIT, no torch extension machinery:
the product is a plain C-ABI shared library (include/sudormrf_hip.h)."""
import hashlib
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libsudormrf_hip.so")
SOURCES = ["srf_api.hip", "srf_encoder.hip", "srf_elementwise.hip", "srf_dwconv.hip", "srf_pyramid.hip", "srf_pyramid_reg.hip", "srf_pwconv.hip", "srf_pwconv_bf16x3.hip", "srf_pwconv_x3w.hip", "srf_pwconv_x3p.hip", "srf_pwconv_x3f.hip", "srf_pwconv_w4.hip", "srf_pwconv_small.hip",
           "srf_tac.hip", "srf_loss.hip", "srf_pwconv_wgrad.hip", "srf_backward.hip", "srf_train.hip", "srf_augment.hip", "srf_optim.hip", "srf_feeder.hip"]
HEADERS = [os.path.join(CSRC, "srf_common.h"), os.path.join(CSRC, "srf_pw.h"), os.path.join(CSRC, "srf_plan.h"), os.path.join(CSRC, "srf_pyr.h"),
           os.path.join(os.path.dirname(PKG), "include", "sudormrf_hip.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]
# Per-file flags.  The MFMA GEMM files are built without the SLP vectorizer: what it forms there are packed-fp32 VALU
# instructions (v_pk_mul/fma/add_f32) out of the operand prologue's scalar code -- not fewer instructions (the persistent
# res_conv kernel: 1721 VALU with, 1702 without), more expensive beside MFMAs (MI355X_MICROARCH.md, "price of one filler"),
# and the source of the hazardous operand form below.
FILE_FLAGS = {
    "srf_pwconv_bf16x3.hip": ["-fno-slp-vectorize"],
    "srf_pwconv_x3w.hip": ["-fno-slp-vectorize"],
    "srf_pwconv_x3p.hip": ["-fno-slp-vectorize"],
    "srf_pwconv_x3f.hip": ["-fno-slp-vectorize"],
    "srf_pwconv_w4.hip": ["-fno-slp-vectorize"],
    "srf_pwconv_wgrad.hip": ["-fno-slp-vectorize"],
    "srf_pwconv.hip": ["-fno-slp-vectorize"],
    # register-resident pyramid: the SLP vectorizer trades two v_fma_f32 for one v_pk_fma_f32 plus a v_mov that builds the
    # operand pair (same VALU count, dearer instructions: tools/probes/valu_rate_probe.hip measures 4.6 vs 5.5 cycles);
    # same-box A/B on cfg 2: pass 1 70.0 -> 63.7 us, pass 2 unchanged
    "srf_pyramid_reg.hip": ["-fno-slp-vectorize"],
}

# ISA lint (gfx950 erratum found in round 2, tools/probes/pk_opsel_probe.hip): a VOP3P packed-fp32 instruction whose SRC1
# carries op_sel = 1 (the LOW result reads the HIGH dword of src1) returns a wrong low result in lanes 48..63 while another
# wavefront's bf16 MFMA executes on the same SIMD -- silently, and only under co-residency with an MFMA kernel (another
# stream, or the kernel's own other wavefronts).  hipcc emits the form freely (SLP vectorizer, float2 arithmetic with a
# scalar taken from the high half of a loaded pair).  No object containing it is accepted.
_HAZARD = re.compile(r"^\s*(v_pk_\w+)\s.*op_sel:\[[01],1[,\]]", re.M)


def isa_lint(asm_path):
    """[(line number, instruction text)] of hazardous packed instructions in a device assembly file."""
    hits = []
    with open(asm_path, errors="replace") as f:
        for i, line in enumerate(f, 1):
            if _HAZARD.match(line):
                hits.append((i, line.strip()))
    return hits


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def _digest(paths, extra=""):
    h = hashlib.sha1(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _compile(cc, src, extra_flags):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
    stamp = obj + ".sha1"
    flags = FLAGS + FILE_FLAGS.get(src, []) + extra_flags
    dig = _digest([path, os.path.abspath(__file__)] + HEADERS, " ".join(flags))
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    # -save-temps=obj leaves the device assembly next to the object: that is what the ISA lint reads
    cmd = [cc] + flags + ["-save-temps=obj", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    warn = "\n".join(l for l in r.stderr.splitlines() if "argument unused during compilation" not in l)
    if warn.strip():
        sys.stderr.write(warn + "\n")
    base = os.path.basename(src).replace(".hip", "")
    asm = os.path.join(OBJ, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(asm):
        raise RuntimeError("ISA lint: device assembly %s was not produced" % asm)
    hits = isa_lint(asm)
    for f in os.listdir(OBJ):          # drop the temporaries (preprocessed sources, bitcode, fat binaries)
        if f.startswith(base + "-h") or f.startswith(base + ".hip-"):
            os.remove(os.path.join(OBJ, f))
    if hits:
        os.remove(obj)
        raise RuntimeError(
            "ISA lint: %s contains %d packed-fp32 instruction(s) with op_sel = 1 on src1 (wrong low result in lanes 48..63 "
            "next to another wavefront's bf16 MFMA on gfx950, see build.py); first: line %d: %s"
            % (src, len(hits), hits[0][0], hits[0][1]))
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(force=False, verbose=True, extra_flags=()):
    """Compile every HIP source for gfx950 and link the shared library.  Incremental."""
    cc = hipcc()
    os.makedirs(OBJ, exist_ok=True)
    extra_flags = list(extra_flags)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    sources = SOURCES
    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 4)) as ex:
        res = list(ex.map(lambda s: _compile(cc, s, extra_flags), sources))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = [cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
