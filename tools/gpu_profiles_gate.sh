#!/bin/bash
# Runs ON the GPU box: tools/gpu_profiles.sh, but only on a box whose cfg-2 forward is within the pool's usual range (the pool has
# boxes that run the same binary 6-8 % slower at a lower package power; the judged evidence should not come from an outlier).
set -u
export TMPDIR=/tmp
ms=$(timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
echo "gate: cfg-2 forward $ms ms"
python -c "import sys; sys.exit(0 if float('$ms') <= ${GATE_MS:-6.95} else 1)" || { echo "gate: slow box, not profiling"; exit 3; }
bash tools/gpu_profiles.sh "$@"
