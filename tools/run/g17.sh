cd /root/repo
for w in ragged config2; do python bench.py --steps 5 --warmup 2 --main-only --workload $w --f32-first 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['config']['workload'][:30], l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['kernels_per_launch'])"; done
python -m pytest tests/test_f32_first.py tests/test_underflow_band.py tests/test_hip_parity.py -x -q 2>&1 | tail -3
timeout 300 python tools/soak.py 60 11 2>&1 | tail -1 | cut -c1-200
