# forward cfg 2: 2-way auto split against explicit 3- and 4-way splits (SRF_STREAM_SPLIT weights)
for rep in 1 2; do
for sp in auto 1:1:1 3:3:2 2:1:1 1:1:1:1; do
SRF_STREAM_SPLIT=$sp timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$sp', round(d['ms_per_step'],3), d['config'].get('stream_split'), d.get('self_check',{}).get('ok'))"
done; done
