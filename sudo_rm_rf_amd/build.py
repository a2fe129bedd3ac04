"""Build libsudormrf_hip.so (gfx950) in-tree with hipcc.  No JIT, no torch extension machinery:
the product is a plain C-ABI shared library (include/sudormrf_hip.h)."""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libsudormrf_hip.so")
SOURCES = ["srf_api.hip", "srf_encoder.hip", "srf_elementwise.hip", "srf_dwconv.hip", "srf_pyramid.hip", "srf_pyramid_reg.hip", "srf_pwconv.hip", "srf_pwconv_bf16x3.hip", "srf_pwconv_x3p.hip", "srf_pwconv_small.hip",
           "srf_tac.hip", "srf_loss.hip", "srf_pwconv_wgrad.hip", "srf_backward.hip", "srf_train.hip", "srf_augment.hip", "srf_optim.hip"]
HEADERS = [os.path.join(CSRC, "srf_common.h"), os.path.join(CSRC, "srf_pw.h"), os.path.join(CSRC, "srf_plan.h"), os.path.join(CSRC, "srf_pyr.h"),
           os.path.join(os.path.dirname(PKG), "include", "sudormrf_hip.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


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
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha1"
    dig = _digest([path] + HEADERS, " ".join(FLAGS + extra_flags))
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [cc] + FLAGS + extra_flags + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
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
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        res = list(ex.map(lambda s: _compile(cc, s, extra_flags), SOURCES))
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
