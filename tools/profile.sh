#!/bin/bash
# Developer tool: rocprofv3 kernel trace + PMC passes of one bench.py workload on the GPU box (run through gpurun).
# Usage: tools/profile.sh <tag> [bench args...]      e.g. tools/profile.sh r02_config3 --workload config3
# Output: gpurun_out/<tag>/{bench.json, trace/, pmc_*/}; tools/rocpd_summary.py + tools/pmc_update.py turn it into the
# files committed under profiles/ (summary, PMC json, and the entry of profiles/pmc_traffic.json bench.py reads).
set -u
TAG=${1:-r02}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
BA="--steps 20 --warmup 5 --main-only $*"
python bench.py $BA > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
# (tools/rocpd_summary.py reports the mean over the TIMED launches -- the last `steps` of every kernel -- beside the mean over
# all dispatches: the steady state the bench line reports, no warm-up dispatch in it)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py $BA > $OUT/trace.log 2>&1
# (the counter passes keep every dispatch: instruction and byte counts do not depend on clocks or warm caches)
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C -d $OUT/pmc_$N -o pmc -- python bench.py $BA > $OUT/pmc_$N.log 2>&1
done
find $OUT -name "*.db" | head -20
