#!/bin/bash
# gpurun recipe: every entry point of tools/threads_bench under concurrency with TB_VERIFY=1 -- each call's results against the
# same region's first pass (likelihoods within 1e-9, everything discrete equal), C++ threads, no interpreter in the way.
# usage (on the GPU box): bash tools/run/verify_threads.sh <round> [seconds per point]
R=${1:-r04}; S=${2:-2}
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp TB_VERIFY=1 PHMM_MIRROR_CANARY=1
out=gpurun_out/${R}_verify_threads.txt
mkdir -p gpurun_out
echo "# tools/threads_bench TB_VERIFY=1, $S s per point: a line per point; 'TB_VERIFY' lines are failures" > $out
for shape in "128 8 150 300 1" "30 3 150 300 1" "ragged" "128 8 150 300 4"; do
  for m in own shared pipeline fused gshared; do
    for t in 1 2 3 4 8 16; do
      if [ "$shape" = ragged ]; then
        TB_SHAPE=ragged TB_MODE=$m TB_THREADS=$t tools/threads_bench $S 2>&1 | grep "TB_VERIFY\|regions/s\|failed" | sed "s/^/ragged: /" >> $out
      else
        TB_MODE=$m TB_THREADS=$t tools/threads_bench $S $shape 2>&1 | grep "TB_VERIFY\|regions/s\|failed" | sed "s/^/$shape: /" >> $out
      fi
    done
  done
done
echo "points: $(grep -c 'regions/s' $out), failures: $(grep -v '^#' $out | grep -c 'TB_VERIFY\|failed')" | tee -a $out
