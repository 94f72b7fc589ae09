#!/bin/bash
# gpurun recipe: one region per call from T C++ threads -- the resident region server (default) against the launched pipeline
# (PHMM_REGION_SERVER=0), same box, same run.   usage (on the GPU box): bash tools/run/server_rates.sh <round>
R=${1:-r06}
cd "$(dirname "$0")/../.."
O=gpurun_out/${R}_server_rates.txt
{
echo "# tools/threads_bench: phmm_region_compute / phmm_region_submit, one region (128 x 8, 150 / 300) per call, regions/s"
echo "## region server (default): private handles"
TB_MODE=fused TB_THREADS=1,2,4,8,10,16,32 tools/threads_bench 1
echo "## region server: one shared handle"
TB_MODE=gshared TB_THREADS=1,8,10,16,32 tools/threads_bench 1
echo "## region server: one shared handle, two tickets per worker"
TB_DEPTH=2 TB_MODE=gshared TB_THREADS=8,10,16 tools/threads_bench 1
echo "## launched pipeline (PHMM_REGION_SERVER=0): private handles"
PHMM_REGION_SERVER=0 TB_MODE=fused TB_THREADS=1,4,10,16 tools/threads_bench 1
echo "## launched pipeline: one shared handle"
PHMM_REGION_SERVER=0 TB_MODE=gshared TB_THREADS=8,10,16 tools/threads_bench 1
echo "## region server, 30 x 3 regions (R 100 / H 200)"
TB_MODE=fused TB_THREADS=1,10,16 tools/threads_bench 1 30 3 100 200
echo "## region server, ragged mix"
TB_SHAPE=ragged TB_MODE=fused TB_THREADS=1,10,16 tools/threads_bench 1
} > $O 2>&1
cat $O
