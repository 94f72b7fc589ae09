#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_region_handoffs.py -x -q -m gpu -k "not cpp_callers" 2>&1 | tail -25 > gpurun_out/r05_handoffs.log
timeout 900 python -m pytest tests/test_submit_wait.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r05_submit.log
{
TB_MODE=own TB_THREADS=4,5,8,16,32 tools/threads_bench 1.5
TB_MODE=fused TB_THREADS=4,5,8,16,32 tools/threads_bench 1.5
TB_MODE=pipeline TB_THREADS=4,8,16,32 tools/threads_bench 1.5
} > gpurun_out/r05_threads_routed.txt 2>&1
(time timeout 1500 python -m pytest tests/test_region_handoffs.py -q -m gpu -k "cpp_callers") 2>&1 | tail -12 > gpurun_out/r05_verify_matrix.log
cat gpurun_out/r05_handoffs.log gpurun_out/r05_submit.log gpurun_out/r05_threads_routed.txt gpurun_out/r05_verify_matrix.log
