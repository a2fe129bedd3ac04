for rep in 1 2; do
for cfg in "off auto" "on half" "on 9:7" "on 7:9" "on 17:15" "auto auto"; do
set -- $cfg
SRF_PHASE_LOCK=$1 SRF_STREAM_SPLIT=$2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lock=$1 split=$2', round(d['ms_per_step'],3), d['config'].get('stream_split'), d['config'].get('stream_phase_lock'), d.get('self_check',{}).get('ok'))"
done; done
