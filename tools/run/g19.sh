cd /root/repo
for fc in "" 20 12 8 5; do E=""; [ -n "$fc" ] && E="PHMM_FORCE_CHAIN=$fc"; echo "force_chain=$fc"; env $E python bench.py --steps 5 --warmup 2 --main-only --workload ragged 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', l['value'], l['ms_per_step'], l['roofline']['kernels_per_launch'])"; done
