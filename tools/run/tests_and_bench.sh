#!/bin/bash
# gpurun recipe 1: the GPU test suite and the default bench line.   usage (on the GPU box): bash tools/run/tests_and_bench.sh <round>
# -> gpurun_out/<round>_gpu_tests.log, gpurun_out/<round>_bench.json (copy the latter to profiles/<round>_final_bench.json)
R=${1:-r05}
cd "$(dirname "$0")/../.."
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${R}_gpu_tests.log
python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
tail -3 gpurun_out/${R}_gpu_tests.log; tail -c 400 gpurun_out/${R}_bench.json
