#!/bin/bash
# Round 4, first GPU call: is the 256 x 128 GEMM power-bound?   + two cheap structure experiments.
set -u
OUT=gpurun_out/${1:-r04a}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4; nproc; } > "$OUT/env.log" 2>&1
echo "== power probe: full / nomfma / nomfma_noepi / nomfma_noload"
for spec in "proj 0" "proj $((4<<16))" "proj $((20<<16))" "proj $((7<<16))" "proj $((19<<16))" "res_conv 0" "res_conv $((4<<16))" "res_conv $((20<<16))" "res_conv $((23<<16))" "copy 0" "idle 0"; do
  set -- $spec
  timeout 120 python tools/power_probe.py $1 2.5 $2 2>/dev/null | tail -1 | tee -a "$OUT/power_probe.log" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['what'], d['debug_flags']>>16, 'us', round(d['us_per_call'],1), 'W', d['power_w_median'], 'sclk', d['sclk_mhz_median'], 'n', d['samples'])"
done
echo "== gemm A/B: de-phased SIMD partners (256), fake wide loads (512), both (768)"
GEMM_SHAPES=proj_1x1,res_conv GEMM_ROUNDS=7 GEMM_ITERS=20 timeout 600 python tools/gemm_ab.py x3w=0:0 dephase=0:256 wide=0:512 both=0:768 > "$OUT/gemm_ab.log" 2>&1; echo "rc=$?"; grep -v "^{" "$OUT/gemm_ab.log" | tail -12
echo "== done"
