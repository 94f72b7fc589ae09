#!/bin/bash
# gpurun recipe: where a region call's time goes -- inside the resident region server (device timestamps per task, tools/server_trace)
# at 1 / 10 / 16 callers, and the launched pipeline behind the shared handle's combiner at 16 callers under rocprofv3 --kernel-trace
# (round 5's trace, same box).   usage (on the GPU box): bash tools/run/server_trace.sh <round>
R=${1:-r06}
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "# tools/server_trace: T callers x N region calls (128 reads x 8 haplotypes, 150 / 300) through the region server, every task's"
echo "# claim / begin / end and a chain's phases by the device's 100 MHz clock"
for t in 1 10 16; do echo "## $t caller(s)"; timeout 120 tools/server_trace $t 300; done
echo "## 10 callers, 30 reads x 3 haplotypes (100 / 200)"
timeout 120 tools/server_trace 10 300 30 3 100 200
echo
echo "# the launched pipeline behind the shared handle's combiner, 16 callers, rocprofv3 --kernel-trace (round 5's recipe)"
TB_MODE=gshared TB_THREADS=10,16 tools/threads_bench 1.5 | grep gshared
rm -rf /tmp/gs16; PHMM_SUBMIT_STATS=1 TB_MODE=gshared TB_THREADS=16 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gs16 -o gs -- tools/threads_bench 1.0 2>&1 | grep "threads:\|flushes"
python tools/trace_overlap.py $(find /tmp/gs16 -name "*kernel_trace.csv" | head -1) 2>&1 | head -30
python3 - <<'PY'
import csv, glob
for r in list(csv.reader(open(glob.glob("/tmp/gs16/**/*kernel_stats.csv", recursive=True)[0])))[:12]:
    print("%-72s %s" % (r[0][:72], "  ".join("%12s" % x[:12] for x in r[1:6])))
PY
echo
echo "# the region server under rocprofv3 --kernel-trace --stats, 16 callers with a handle each: ONE kernel"
rm -rf /tmp/srv16; TB_MODE=fused TB_THREADS=16 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/srv16 -o srv -- tools/threads_bench 1.0 2>&1 | grep "threads:"
python3 - <<'PY'
import csv, glob
for r in list(csv.reader(open(glob.glob("/tmp/srv16/**/*kernel_stats.csv", recursive=True)[0])))[:6]:
    print("%-72s %s" % (r[0][:72], "  ".join("%12s" % x[:12] for x in r[1:6])))
PY
} > gpurun_out/${R}_server_trace.txt 2>&1
cat gpurun_out/${R}_server_trace.txt
