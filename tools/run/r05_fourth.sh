#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "## gshared vs fused (routed), same box"
TB_MODE=gshared TB_THREADS=8,10,16,32 tools/threads_bench 1.5
TB_MODE=fused TB_THREADS=8,10,16,32 tools/threads_bench 1.5
PHMM_SUBMIT_STATS=1 TB_MODE=fused TB_THREADS=16 tools/threads_bench 1.5
PHMM_SUBMIT_STATS=1 TB_MODE=gshared TB_THREADS=16 tools/threads_bench 1.5
for l in 4 6 8; do for d in 1 2; do echo "lanes $l depth $d"; PHMM_SUBMIT_LANES=$l TB_DEPTH=$d TB_MODE=gshared TB_THREADS=8,10,16,32 tools/threads_bench 1.5 | grep gshared; done; done
echo "## gather 0"
PHMM_SUBMIT_GATHER_US=0 TB_MODE=gshared TB_THREADS=8,10,16 tools/threads_bench 1.5 | grep gshared
PHMM_SUBMIT_GATHER_US=0 TB_DEPTH=2 TB_MODE=gshared TB_THREADS=8,10,16 tools/threads_bench 1.5 | grep gshared
} > gpurun_out/r05_threads_lanes.txt 2>&1
cat gpurun_out/r05_threads_lanes.txt
