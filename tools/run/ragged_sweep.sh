#!/bin/bash
# gpurun recipe: 1 536 mixed regions through host buffers by chunk schedule (tools/hostpath_ragged_sweep.py), every schedule twice
# on one box -> gpurun_out/r05_ragged_sweep2.txt (profiles/r05_ragged_chunk_sweep.txt)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2; do
for fc in "4096 32768" "4096 8192" "4096 6144" "4096 12288" "3072 8192" "6144 8192"; do set -- $fc; PHMM_MIXED_FIRST_CHUNK_KB=$1 PHMM_MIXED_CHUNK_KB=$2 python tools/hostpath_ragged_sweep.py; done
for fc in "1024 8192" "2048 8192" "1024 12288" "2048 6144" "512 8192"; do set -- $fc; echo -n "flat: "; PHMM_MIXED_FLAT=1 PHMM_MIXED_FIRST_CHUNK_KB=$1 PHMM_MIXED_CHUNK_KB=$2 python tools/hostpath_ragged_sweep.py; done
done
} > gpurun_out/r05_ragged_sweep2.txt 2>&1
cat gpurun_out/r05_ragged_sweep2.txt
