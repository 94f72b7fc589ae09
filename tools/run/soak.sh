#!/bin/bash
# gpurun recipe 4: random batches of every kind against the oracle (PairHMM through every kernel selection, the
# engine-level call, Smith-Waterman, the projection).  usage (on the GPU box): bash tools/run/soak.sh <round> [seconds each]
R=${1:-r05}; S=${2:-150}
cd "$(dirname "$0")/../.."
export PHMM_MIRROR_CANARY=1   # a device store that lands in the pinned mirror outside its call fails that call (phmm_api.cpp)
{
timeout $((S * 2 + 100)) python tools/soak.py $S 71 2>&1 | tail -1
timeout $((S * 2 + 100)) python tools/soak_engine.py $S 72 2>&1 | tail -1
timeout $((S * 2 + 100)) python tools/soak_sw.py $S 73 2>&1 | tail -1
timeout $((S * 2 + 100)) python tools/soak_project.py $S 74 2>&1 | tail -1
timeout $((S * 2 + 100)) python tools/soak_region.py $S 75 2>&1 | tail -1
timeout $((S * 2 + 100)) python tools/soak_region_threads.py $S 4 76 2>&1 | tail -1
} > gpurun_out/${R}_soak_parity.txt 2>&1
cat gpurun_out/${R}_soak_parity.txt
