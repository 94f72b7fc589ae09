cd /root/repo
cp lorikeet_amd/libphmm.so /tmp/cur.so
for rep in 1 2; do for v in DPHMM_MIXED_RUNS64 DPHMM_MIXED_RUNS64DPHMM_NO_COST_SCALE DPHMM_MIXED_RUNS32DPHMM_NO_COST_SCALE DPHMM_MIXED_RUNS48DPHMM_NO_COST_SCALE; do cp tools/ab/libphmm_$v.so lorikeet_amd/libphmm.so; echo -n "$v: "; python bench.py --steps 6 --warmup 2 --main-only --workload ragged 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'])"; done; done
cp /tmp/cur.so lorikeet_amd/libphmm.so
