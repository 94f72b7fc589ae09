#!/bin/bash
# gpurun recipe: the shared handle's region call at 16 caller threads under rocprofv3 --kernel-trace: kernels in flight, per-kernel
# durations under load; and the tags-only aligner on / off on the same box
R=${1:-r05}
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "## A/B on one box: PHMM_SW_LITE default / 0"
for rep in 1 2; do
TB_MODE=gshared TB_THREADS=8,16 tools/threads_bench 1.5 | grep gshared
PHMM_SW_LITE=0 TB_MODE=gshared TB_THREADS=8,16 tools/threads_bench 1.5 | grep gshared | sed 's/^/lite off: /'
done
echo "## 16 threads, kernel trace"
rm -rf /tmp/gs16; PHMM_SUBMIT_STATS=1 TB_MODE=gshared TB_THREADS=16 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gs16 -o gs -- tools/threads_bench 1.0 2>&1 | grep "threads:\|flushes"
python tools/trace_overlap.py $(find /tmp/gs16 -name "*kernel_trace.csv" | head -1) 2>&1 | head -30
python3 - <<'PY'
import csv, glob
for r in list(csv.reader(open(glob.glob("/tmp/gs16/**/*kernel_stats.csv", recursive=True)[0])))[:14]:
    print("%-72s %s" % (r[0][:72], "  ".join("%12s" % x[:12] for x in r[1:6])))
PY
} > gpurun_out/${R}_gshared_trace.txt 2>&1
cat gpurun_out/${R}_gshared_trace.txt
