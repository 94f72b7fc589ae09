cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r02_ragged
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_ragged/trace -o trace -- python bench.py --steps 3 --warmup 1 --main-only --workload ragged > gpurun_out/r02_ragged/trace.log 2>&1
tail -3 gpurun_out/r02_ragged/trace.log | cut -c1-400
python - <<'PY'
import sqlite3, glob
db=glob.glob("gpurun_out/r02_ragged/trace/*.db")[0]
c=sqlite3.connect(db).cursor()
print([r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")][:60])
PY
