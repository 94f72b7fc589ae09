cd /root/repo
for fc in -1 4 6 8 12 16 24; do echo -n "force_chain=$fc: "; PHMM_FORCE_CHAIN=$fc python bench.py --steps 6 --warmup 2 --main-only --workload ragged 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'])"; done
