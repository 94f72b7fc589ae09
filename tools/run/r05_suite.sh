#!/bin/bash
# the whole GPU suite and the default bench line
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > gpurun_out/r05_gpu_tests.log 2>&1
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -30 gpurun_out/r05_gpu_tests.log; tail -c 4200 gpurun_out/r05_bench.json; tail -5 gpurun_out/r05_bench.err
