#!/bin/bash
# gpurun recipe: the call-pattern table, the soaks (canary on), the C++ verify matrix, the region timeline -- one call on the round's
# final code.  usage (on the GPU box): bash tools/run/call_pattern.sh   -> gpurun_out/r05_{threads_bench,soak_parity,verify_threads,region_timeline}.txt
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/run/threads.sh r05 > /dev/null 2>&1
bash tools/run/soak.sh r05 60 > /dev/null 2>&1
bash tools/run/verify_threads.sh r05 3 > /dev/null 2>&1
bash tools/run/timeline.sh r05 > /dev/null 2>&1
tail -3 gpurun_out/r05_threads_bench.txt; cat gpurun_out/r05_soak_parity.txt; tail -2 gpurun_out/r05_verify_threads.txt; tail -5 gpurun_out/r05_region_timeline.txt
