#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests/test_sw_hip.py tests/test_realign_hip.py tests/test_region_hip.py tests/test_calculate_cigar_hip.py tests/test_project_hip.py -m gpu -q -x 2>&1 | tail -30) > gpurun_out/r05_sw_tests.log 2>&1
timeout 300 python tools/soak_sw.py 60 73 2>&1 | tail -3 >> gpurun_out/r05_sw_tests.log
timeout 300 python tools/soak_region.py 60 75 2>&1 | tail -2 >> gpurun_out/r05_sw_tests.log
python tools/sw_bench.py 2>&1 | tail -8 >> gpurun_out/r05_sw_tests.log
cat gpurun_out/r05_sw_tests.log
