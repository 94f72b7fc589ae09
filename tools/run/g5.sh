cd /root/repo
python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(l['realign_to_best'])[:1800])"
