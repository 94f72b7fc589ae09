#!/bin/bash
# gpurun recipe: one region per call from T C++ threads -- the resident region server against the launched pipeline and the
# shared handle's combiner, same box, same run.   usage (on the GPU box): bash tools/run/server_rates.sh <round>
R=${1:-r06}
cd "$(dirname "$0")/../.."
O=gpurun_out/${R}_server_rates.txt
{
echo "# tools/threads_bench: phmm_region_compute / phmm_region_submit, one region (128 x 8, 150 / 300) per call, regions/s"
echo "## private handles, defaults (up to five: their own launched pipelines; past five: the region server)"
TB_MODE=fused TB_THREADS=1,2,4,5,6,8,10,16,32 tools/threads_bench 1
echo "## private handles, every call through the region server (PHMM_REGION_SERVER=1)"
PHMM_REGION_SERVER=1 TB_MODE=fused TB_THREADS=1,4 tools/threads_bench 1
echo "## private handles, launched pipeline only (PHMM_REGION_SERVER=0)"
PHMM_REGION_SERVER=0 TB_MODE=fused TB_THREADS=8,10,16,32 tools/threads_bench 1
echo "## private handles, launched pipeline routed through the combiner past four (PHMM_REGION_SERVER=0 PHMM_ROUTE_SHARED=4: round 5's default)"
PHMM_REGION_SERVER=0 PHMM_ROUTE_SHARED=4 TB_MODE=fused TB_THREADS=8,10,16,32 tools/threads_bench 1
echo "## one shared handle (phmm_region_submit / phmm_wait: the combiner)"
TB_MODE=gshared TB_THREADS=8,10,16,32 tools/threads_bench 1
echo "## one shared handle through the region server (PHMM_REGION_SERVER=1)"
PHMM_REGION_SERVER=1 TB_MODE=gshared TB_THREADS=8,10,16,32 tools/threads_bench 1
echo "## ... two tickets per worker"
PHMM_REGION_SERVER=1 TB_DEPTH=2 TB_MODE=gshared TB_THREADS=8,10,16 tools/threads_bench 1
echo "## region server, 30 x 3 regions (R 100 / H 200), private handles"
TB_MODE=fused TB_THREADS=10,16 tools/threads_bench 1 30 3 100 200
echo "## region server, ragged mix, private handles"
TB_SHAPE=ragged TB_MODE=fused TB_THREADS=10,16 tools/threads_bench 1
echo "## ... launched pipeline"
PHMM_REGION_SERVER=0 TB_SHAPE=ragged TB_MODE=fused TB_THREADS=10,16 tools/threads_bench 1
} > $O 2>&1
cat $O
