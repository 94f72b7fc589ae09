#!/bin/bash
# gpurun recipe 3: the reference's call pattern (one region per call from T host threads) through every entry point.
# usage (on the GPU box): bash tools/run/threads.sh <round>   -> gpurun_out/<round>_threads_bench.txt (copy to profiles/)
R=${1:-r05}
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/${R}_threads_bench.txt
{
echo "# tools/threads_bench on MI355X: one region per call from T C++ host threads, host buffers (PCIe included)"
echo "## PairHMM alone (phmm_compute / phmm_submit), private handles and one shared handle"
TB_THREADS=1,2,4,8,10,16,32 tools/threads_bench 1
echo "## likelihoods, then realignment, two calls per region (phmm_compute, then phmm_realign_reads with its likelihoods)"
TB_MODE=pipeline TB_THREADS=1,2,4,8,16 tools/threads_bench 1
echo "## the whole per-region path as ONE call (phmm_region_compute), private handles (past five: through the resident region server)"
TB_MODE=fused TB_THREADS=1,2,4,5,8,10,16,32 tools/threads_bench 1
echo "## ... through the shared handle (phmm_region_submit / phmm_wait)"
TB_MODE=gshared TB_THREADS=1,2,4,8,10,16,32,64 tools/threads_bench 1
echo "## ... two tickets in flight per worker (TB_DEPTH=2: region k+1 submitted before region k is waited for)"
TB_DEPTH=2 TB_MODE=gshared TB_THREADS=4,8,10,16,32 tools/threads_bench 1
echo "## ... private handles with the routing off (PHMM_ROUTE_SHARED=0: round 4's behaviour)"
PHMM_ROUTE_SHARED=0 TB_MODE=fused TB_THREADS=8,16,32 tools/threads_bench 1
PHMM_ROUTE_SHARED=0 TB_MODE=own TB_THREADS=8,16,32 tools/threads_bench 1
echo "## ... by regions per call (one and four caller threads)"
for pc in 2 4 8 16 64 256; do TB_MODE=fused TB_THREADS=1,4 tools/threads_bench 0.7 128 8 150 300 $pc | grep fused | sed "s/^/$pc regions per call: /"; done
echo "## kernels of one region call (rocprofv3 --kernel-trace --stats, one caller thread)"
rm -rf gpurun_out/tb1; mkdir -p gpurun_out/tb1
TB_MODE=fused TB_THREADS=1 rocprofv3 --kernel-trace --stats -d gpurun_out/tb1 -o tb1 --output-format csv -- tools/threads_bench 0.4 2>&1 | grep threads:
python3 - <<'PY'
import csv, glob
for r in list(csv.reader(open(glob.glob("gpurun_out/tb1/**/*kernel_stats.csv", recursive=True)[0])))[:12]:
    print("%-72s %s" % (r[0][:72], "  ".join("%12s" % x[:12] for x in r[1:6])))
PY
rm -rf gpurun_out/tb1
} > $O 2>&1
tail -5 $O
