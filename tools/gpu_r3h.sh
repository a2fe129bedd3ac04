#!/bin/bash
set -u
OUT=gpurun_out/${1:-r03h}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
c() { echo $(( ($1 + 1) << 26 )); }
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:24], "ms", round(d["ms_per_step"], 3), "value", round(d["value"]), "frac", round(d["forward_roofline"]["frac"], 3), {k[-6:]: round(v["avg_launch_us"], 1) for k, v in d.get("kernels", {}).items() if "x3v" in k or "pyr" in k})
PY
}
for rep in 1 2; do
for f in ${FLAGS:-0 $((1<<24)) 32768 $(c 0)}; do
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --debug-flags $f > "$OUT/bench_f$f.json" 2> "$OUT/bench_f$f.err"; echo -n "flags $f: "; show "$OUT/bench_f$f.json"
done; done
echo "== done"
