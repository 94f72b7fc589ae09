#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2 3; do
echo -n "fork last: "; PHMM_FORK_LAST_CHUNK=1 python tools/hostpath_ragged_sweep.py 2>&1 | grep -v amdgpu
echo -n "no fork:   "; PHMM_FORK_LAST_CHUNK=0 python tools/hostpath_ragged_sweep.py 2>&1 | grep -v amdgpu
done
python -m pytest tests/test_full_size_configs.py tests/test_hip_parity.py tests/test_engine_hip.py -m gpu -q -x 2>&1 | tail -3
} > gpurun_out/r05_fork_last.txt 2>&1
cat gpurun_out/r05_fork_last.txt
