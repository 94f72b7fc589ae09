#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
python -m pytest tests/test_sw_hip.py tests/test_realign_hip.py tests/test_region_hip.py tests/test_calculate_cigar_hip.py -m gpu -q -x 2>&1 | tail -3
python tools/sw_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
python tools/sw_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 200 python tools/soak_sw.py 40 73 2>&1 | tail -1
timeout 200 python tools/soak_region.py 30 75 2>&1 | tail -1
TB_MODE=gshared TB_THREADS=8,16 tools/threads_bench 1.5
TB_MODE=fused TB_THREADS=1 tools/threads_bench 1.5
} > gpurun_out/r05_swb.txt 2>&1
cat gpurun_out/r05_swb.txt
