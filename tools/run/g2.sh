set -x
cd /root/repo
python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; tail -c 300 gpurun_out/bench_r02a.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/bench_r02a.json").read().strip().splitlines()[-1])
for k,v in l.items(): print(k, json.dumps(v)[:700])
PY
