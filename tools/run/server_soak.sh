#!/bin/bash
# gpurun recipe: the region server under C++ callers that verify every call against their first pass (tools/threads_bench
# TB_VERIFY=1, the mirror canary on), long points.   usage (on the GPU box): bash tools/run/server_soak.sh <round> [seconds per point]
R=${1:-r06}; S=${2:-30}
cd "$(dirname "$0")/../.."
O=gpurun_out/${R}_server_soak.txt
{
echo "# tools/threads_bench TB_VERIFY=1 PHMM_MIRROR_CANARY=1, $S s per point: the region calls of private handles past five go through the region server"
for shape in ragged config2; do
  for mode in fused; do
    echo "## $mode, $shape"
    if [ $shape = ragged ]; then export TB_SHAPE=ragged; else unset TB_SHAPE; fi
    PHMM_MIRROR_CANARY=1 TB_VERIFY=1 TB_MODE=$mode TB_THREADS=8,16,32 timeout $((S * 4 + 60)) tools/threads_bench $S 2>&1 | tail -5
  done
done
echo "## shared handle through the server (PHMM_REGION_SERVER=1), ragged, two tickets"
PHMM_REGION_SERVER=1 PHMM_MIRROR_CANARY=1 TB_VERIFY=1 TB_SHAPE=ragged TB_DEPTH=2 TB_MODE=gshared TB_THREADS=8,16 timeout $((S * 3 + 60)) tools/threads_bench $S 2>&1 | tail -4
} > $O 2>&1
cat $O
