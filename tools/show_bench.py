#!/usr/bin/env python3
import json, sys
b = json.load(open(sys.argv[1]))
print("value %.0f sep-s/s  %.3f ms/step  hbm-roofline frac %.3f" % (b["value"], b["ms_per_step"], b["forward_roofline"]["frac"]))
for k, v in b.get("kernels", {}).items():
    print("  %-22s %7.3f ms/fwd  x%-3d %8.1f us  %6.0f GB/s  %6.1f TF" % (k, v["ms_per_forward"], v["launches_per_forward"], v["avg_launch_us"], v["algorithmic_GBps"], v["TFLOPs"]))
if "cpu_baseline" in b:
    print("  cpu_baseline", b["cpu_baseline"].get("value"), b["cpu_baseline"].get("cores"))
