#!/bin/bash
# gpurun recipe (round 5, first pass): the new hand-off tests, the region tests, then the call-pattern baseline at Lorikeet's thread counts
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_region_handoffs.py -x -q -m gpu -k "not cpp_callers" 2>&1 | tail -25 > gpurun_out/r05_handoffs.log
timeout 900 python -m pytest tests/test_region_hip.py tests/test_submit_wait.py tests/test_multi.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_region.log
{
TB_MODE=gshared TB_THREADS=8,10,16,32 tools/threads_bench 1.5
TB_MODE=fused TB_THREADS=1,4,8,16,32 tools/threads_bench 1.5
TB_MODE=own TB_THREADS=8,16,32 tools/threads_bench 1.5
TB_MODE=shared TB_THREADS=8,10,16,32 tools/threads_bench 1.5
} > gpurun_out/r05_threads_first.txt 2>&1
cat gpurun_out/r05_handoffs.log gpurun_out/r05_region.log gpurun_out/r05_threads_first.txt
