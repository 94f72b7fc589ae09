#!/bin/bash
# gpurun recipe: long runs of the checks that found round 4's race -- the C++ verify matrix at 10 s per point and the soaks at
# 240 s each, mirror canary on
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/run/verify_threads.sh r05long 10 > /dev/null 2>&1
bash tools/run/soak.sh r05long 240 > /dev/null 2>&1
tail -2 gpurun_out/r05long_verify_threads.txt; cat gpurun_out/r05long_soak_parity.txt
