#!/bin/bash
set -u
OUT=gpurun_out/r05n
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( cd /tmp && SRF_STREAM_SPLIT=off timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d "$GRAFT_REPO_ROOT/$OUT/pmc" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --workload cfg3_groupcomm_u8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile ) > "$OUT/pmc.log" 2>&1
f=$(find "$OUT/pmc" -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0]
    if "tac" not in k and "pw_small" not in k: continue
    e=agg[k][r["Counter_Name"]]; e[0]+=1; e[1]+=float(r["Counter_Value"])
for k,cs in agg.items():
    v={c:x[1]/x[0] for c,x in cs.items()}
    cyc=v.get("SQ_BUSY_CYCLES",0)/32
    print(k[-40:], "n=%d"%max(x[0] for x in cs.values()))
    print("   kernel cycles %.0fk  MFMA busy %.1f%%  MFMA insts %.3g  VALU/SIMD %.1fk  wave_cycles %.3g  wait_any %.1f%%  wait_inst_any %.1f%%  active_inst %.1f%%" % (
        cyc/1e3, 100*v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(1024*cyc) if cyc else 0, v.get("SQ_INSTS_MFMA",0), v.get("SQ_INSTS_VALU",0)/1024e3, v.get("SQ_WAVE_CYCLES",0),
        100*v.get("SQ_WAIT_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1), 100*v.get("SQ_WAIT_INST_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1), 100*v.get("SQ_ACTIVE_INST_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1)))
PY
find "$OUT" -name "*.csv" -size +2M -delete
