set -x
cd /root/repo
python bench.py > gpurun_out/bench_r02e.json 2> gpurun_out/bench_r02e.err; tail -c 300 gpurun_out/bench_r02e.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/bench_r02e.json").read().strip().splitlines()[-1])
for k in ("value","ragged","smith_waterman","single_region"): print(k, json.dumps(l[k])[:1500])
PY
