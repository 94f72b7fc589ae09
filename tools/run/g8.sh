cd /root/repo
for w in config2 config5 ragged; do python bench.py --steps 8 --warmup 2 --main-only --workload $w 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['config']['workload'][:30], l['value'], l['ms_per_step'], l['roofline']['kernel'])"; done
python -m pytest tests/test_hip_parity.py tests/test_underflow_band.py -x -q 2>&1 | tail -3
