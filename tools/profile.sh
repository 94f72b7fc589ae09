#!/bin/bash
# Developer tool: bench + rocprofv3 kernel trace + PMC passes on the GPU box (run through gpurun).
# Usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
BA="--steps 5 --warmup 2 --main-only $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py $BA > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C -d $OUT/pmc_$N -o pmc -- python bench.py $BA > $OUT/pmc_$N.log 2>&1
done
find $OUT -name "*.csv" | head -50
