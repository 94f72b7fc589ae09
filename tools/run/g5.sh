cd /root/repo
python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['realign_to_best']; print(r['ms_per_call'], r['project_to_reference']['ms_per_call'], json.dumps(r.get('realign_reads_one_call')), r.get('error'))"
