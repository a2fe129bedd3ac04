#!/bin/bash
# Runs ON the GPU box: where do the ~58 small device copies per training step come from?  (kernel trace: neighbours of every copy)
set -u
OUT=gpurun_out/${1:-r05l}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/tr" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --train --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile ) > "$OUT/tr.log" 2>&1
f=$(find "$OUT/tr" -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
# last step: from the last "srf_x3w_pack_kernel<true>" to the end
starts=[i for i,r in enumerate(rows) if "srf_x3w_pack_kernel<true>" in r[2]]
seg=rows[starts[-2]:starts[-1]]
cnt=collections.Counter()
for i,r in enumerate(seg):
    if "copyBuffer" in r[2] and "Rect" not in r[2]:
        prev=next((seg[j][2] for j in range(i-1,-1,-1) if "copyBuffer" not in seg[j][2]), "-")
        nxt=next((seg[j][2] for j in range(i+1,len(seg)) if "copyBuffer" not in seg[j][2]), "-")
        cnt[(prev,nxt)]+=1
print("copies in one step:", sum(cnt.values()))
for (p,n),c in cnt.most_common(12): print("  %3d x  after %-45s before %s"%(c,p,n))
PY
find "$OUT" -name "*.csv" -size +1M -delete
