#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_region_handoffs.py -x -q -m gpu -k "not cpp_callers" 2>&1 | tail -25 > gpurun_out/r05_handoffs.log
{
for d in 1 2; do TB_DEPTH=$d TB_MODE=gshared TB_THREADS=4,8,10,16 tools/threads_bench 1.5; done
for l in 2 3 6 8; do echo "lanes $l"; PHMM_SUBMIT_LANES=$l TB_MODE=gshared TB_THREADS=8,16 tools/threads_bench 1.5 | grep gshared; done
TB_DEPTH=2 TB_MODE=shared TB_THREADS=4,8,10,16 tools/threads_bench 1.5
} > gpurun_out/r05_threads_depth.txt 2>&1
(time timeout 1200 python -m pytest tests/test_region_handoffs.py -x -q -m gpu -k "cpp_callers") 2>&1 | tail -12 > gpurun_out/r05_verify_matrix.log
cat gpurun_out/r05_handoffs.log gpurun_out/r05_threads_depth.txt gpurun_out/r05_verify_matrix.log
