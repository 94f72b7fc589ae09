#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
S="30 3 150 300 1"
{
echo "## 30x3 regions"
TB_MODE=gshared TB_THREADS=1,2,4,8,16,32 tools/threads_bench 1 $S | grep gshared
echo "gather 0:"; PHMM_SUBMIT_GATHER_US=0 TB_MODE=gshared TB_THREADS=4,8,16 tools/threads_bench 1 $S | grep gshared
echo "lanes 8:"; PHMM_SUBMIT_LANES=8 TB_MODE=gshared TB_THREADS=8,16,32 tools/threads_bench 1 $S | grep gshared
echo "lanes 8 gather 0:"; PHMM_SUBMIT_GATHER_US=0 PHMM_SUBMIT_LANES=8 TB_MODE=gshared TB_THREADS=8,16,32 tools/threads_bench 1 $S | grep gshared
echo "depth 2:"; TB_DEPTH=2 TB_MODE=gshared TB_THREADS=4,8,16 tools/threads_bench 1 $S | grep gshared
echo "own, routing off:"; PHMM_ROUTE_SHARED=0 TB_MODE=fused TB_THREADS=2,4,8,16 tools/threads_bench 1 $S | grep fused
echo "sw_all forced 1024 pairs:"; PHMM_REGION_SW_ALL=1024 TB_MODE=gshared TB_THREADS=4,8,16 tools/threads_bench 1 $S | grep gshared
PHMM_SUBMIT_STATS=1 TB_MODE=gshared TB_THREADS=8 tools/threads_bench 1 $S 2>&1 | grep "gshared\|flushes"
} > gpurun_out/r05_small.txt 2>&1
cat gpurun_out/r05_small.txt
