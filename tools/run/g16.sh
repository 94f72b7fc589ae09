cd /root/repo
export TMPDIR=/tmp
for V in 0 1; do
for W in config2 config3; do
O=gpurun_out/r02_fetch_${W}_$V; rm -rf $O; mkdir -p $O
E=""; [ $V = 1 ] && E="PHMM_NO_XCD_INTERLEAVE=1"
env $E rocprofv3 --pmc FETCH_SIZE -d $O/pmc_F -o pmc -- python bench.py --steps 3 --warmup 1 --main-only --workload $W > $O/log 2>&1
env $E rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_T -o pmc -- python bench.py --steps 3 --warmup 1 --main-only --workload $W > $O/log2 2>&1
echo "$W no_interleave=$V"; tail -1 $O/log | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'])"
python - <<PY
import sqlite3, glob
for f in glob.glob("$O/pmc_*/*.db"):
    c=sqlite3.connect(f).cursor()
    for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%phmm_forward%' group by counter_name"): print("   ", r)
PY
done; done
