cd /root/repo
BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/bench_n2.json 2> gpurun_out/bench_n2.err
tail -5 gpurun_out/bench_n2.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/bench_n2.json").read().strip().splitlines()[-1])
for k in ("value","n_gpus","ms_per_step","scaling","config3_10k","config5_256","ragged"): print(k, json.dumps(l.get(k))[:900])
PY
