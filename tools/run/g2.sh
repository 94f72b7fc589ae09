set -x
cd /root/repo
python bench.py > gpurun_out/bench_r02f.json 2> gpurun_out/bench_r02f.err; tail -c 300 gpurun_out/bench_r02f.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/bench_r02f.json").read().strip().splitlines()[-1])
for k in ("value","ragged","smith_waterman","single_region"): print(k, json.dumps(l[k])[:1500])
PY
