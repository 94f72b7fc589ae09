#!/bin/bash
# gpurun recipe: where the region server overtakes private handles' own launched pipelines -- 5..8 threads, both ways, three shapes.
# usage (on the GPU box): bash tools/run/server_threshold.sh <round>
R=${1:-r06}
cd "$(dirname "$0")/../.."
O=gpurun_out/${R}_server_threshold.txt
{
echo "# tools/threads_bench, private handles, 5..8 threads: launched pipeline (PHMM_REGION_SERVER=0) against the server (PHMM_REGION_SERVER=1)"
for S in "1" "1 30 3 100 200"; do
  echo "## launched, $S"; PHMM_REGION_SERVER=0 TB_MODE=fused TB_THREADS=4,5,6,7,8 tools/threads_bench $S
  echo "## server, $S";   PHMM_REGION_SERVER=1 TB_MODE=fused TB_THREADS=4,5,6,7,8 tools/threads_bench $S
done
echo "## launched, ragged"; PHMM_REGION_SERVER=0 TB_SHAPE=ragged TB_MODE=fused TB_THREADS=4,5,6,7,8 tools/threads_bench 1
echo "## server, ragged";   PHMM_REGION_SERVER=1 TB_SHAPE=ragged TB_MODE=fused TB_THREADS=4,5,6,7,8 tools/threads_bench 1
} > $O 2>&1
cat $O
