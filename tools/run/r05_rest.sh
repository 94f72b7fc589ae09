#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests/test_bench_contract.py tests/test_region_hip.py tests/test_share_prefixes_hip.py tests/test_submit_wait.py tests/test_sw_hip.py tests/test_underflow_band.py tests/test_realign_hip.py tests/test_project_hip.py -m gpu -q 2>&1 | tail -30) > gpurun_out/r05_gpu_tests_rest.log 2>&1
cat gpurun_out/r05_gpu_tests_rest.log
