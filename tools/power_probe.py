#!/usr/bin/env python3
"""Is a kernel power-bound?  Runs one GEMM form (or the whole cfg-2 forward) back to back for a few seconds while a thread samples
`rocm-smi` (socket power, shader / memory clocks); prints the samples' median.  Usage: power_probe.py [proj|res_conv|forward|copy|idle] [seconds] [debug flags]"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sudo_rm_rf_amd import ops  # noqa: E402

DEV = "cuda:0"


def sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            rec = {}
            for k, v in card.items():
                kl = k.lower()
                if "power" in kl and "(w)" in kl:
                    rec["power_w"] = float(v)
                if kl.startswith("sclk clock speed"):
                    m = re.search(r"(\d+)", str(v))
                    rec["sclk_mhz"] = float(m.group(1)) if m else None
                if kl.startswith("mclk clock speed"):
                    m = re.search(r"(\d+)", str(v))
                    rec["mclk_mhz"] = float(m.group(1)) if m else None
            out.append(rec)
        except Exception as e:  # noqa: BLE001
            out.append({"error": str(e)[:80]})
        time.sleep(0.05)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "proj"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    ops.set_debug_flags(flags)
    g = torch.Generator(device=DEV).manual_seed(0)
    if what in ("proj", "res_conv"):
        Bt, Cin, Cout, L, pro = (32, 256, 512, 3200, 0) if what == "proj" else (32, 512, 256, 3200, 2)
        x = torch.randn(Bt, Cin, L, generator=g, device=DEV)
        w = torch.randn(Cout, Cin, 1, generator=g, device=DEV) * Cin ** -0.5
        bias = torch.randn(Cout, generator=g, device=DEV)
        kw = dict(packed=ops.pack_pw_weight(w))
        if pro == 2:
            kw.update(in_sums=ops.gln_stats(x, Bt), in_gamma=torch.rand(Cin, generator=g, device=DEV) + 0.5,
                      in_beta=torch.randn(Cin, generator=g, device=DEV), in_prelu=torch.tensor([0.25], device=DEV),
                      residual=torch.randn(Bt, Cout, L, generator=g, device=DEV))
        else:
            kw.update(out_sums=ops.new_sums(Bt, DEV))
        fn = lambda: ops.pw_conv(x, w, bias, **kw)
    elif what == "copy":
        a = torch.randn(64 << 20, device=DEV)
        b = torch.empty_like(a)
        fn = lambda: b.copy_(a)
    elif what == "idle":
        fn = lambda: time.sleep(0.001)
    else:
        import bench
        import sudo_rm_rf.dnn.models.improved_sudormrf as improved_sudormrf
        variant, kwm, T, fs, batch = bench.WORKLOADS["cfg2_improved_u16"]
        model = improved_sudormrf.SuDORMRF(**kwm).to(DEV).eval()
        wav = torch.randn(batch, 1, T, device=DEV)
        model._engine().multi_stream = False

        def fn():
            with torch.no_grad():
                model(wav)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, samples))
    th.start()
    t0, n = time.perf_counter(), 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ok = [s for s in samples if "power_w" in s]

    def med(k):
        vals = sorted(s[k] for s in ok if s.get(k) is not None)
        return vals[len(vals) // 2] if vals else None

    print(json.dumps({"what": what, "debug_flags": flags, "us_per_call": e0.elapsed_time(e1) * 1e3 / max(n, 1), "samples": len(ok),
                      "power_w_median": med("power_w"), "sclk_mhz_median": med("sclk_mhz"), "mclk_mhz_median": med("mclk_mhz"),
                      "first_samples": samples[:3]}))


if __name__ == "__main__":
    main()
