cd /root/repo
export BENCH_DIST_BACKEND=gloo
time (timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_n2.json 2> gpurun_out/bench_n2.err)
tail -c 400 gpurun_out/bench_n2.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/bench_n2.json").read().strip().splitlines()[-1])
print(l["n_gpus"], l["value"], l["ms_per_step"], l["scaling"])
for k in ("config3_10k","config5_256"):
    r=l[k]; print(k, r.get("gcups"), r.get("per_rank_cells"), r.get("imbalance"), r.get("error"))
print([k for k in l.keys()])
PY
