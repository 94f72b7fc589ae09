#!/bin/bash
# gpurun recipe 2: kernel trace + PMC passes of every bench workload and of the engine / Smith-Waterman drivers.
# usage (on the GPU box): bash tools/run/profiles.sh <round> [workloads...]   default: all
# then, back in the build container: bash tools/run/profiles_post.sh <round>   (rocpd databases -> profiles/<round>_*)
R=${1:-r05}; shift || true
W=${*:-config2_f64 config2_f32 config3 config5 ragged engine_call sw_bench}
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
COUNTERS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE")
for T in $W; do
  case $T in
    config2_f64) bash tools/profile.sh ${R}_$T --workload config2 > gpurun_out/prof_$T.log 2>&1 ;;
    config2_f32) bash tools/profile.sh ${R}_$T --workload config2 --f32-first > gpurun_out/prof_$T.log 2>&1 ;;
    config3|config5|ragged) bash tools/profile.sh ${R}_$T --workload $T > gpurun_out/prof_$T.log 2>&1 ;;
    engine_call|sw_bench)   # the tools that drive the engine-level call and the aligner (counters: separate passes, --kernel-trace only)
      O=gpurun_out/${R}_$T; rm -rf $O; mkdir -p $O
      A=256; [ $T = sw_bench ] && A=1024
      python tools/$T.py $A > $O/bench.txt 2>&1
      rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python tools/$T.py $A > $O/trace.log 2>&1
      for C in "${COUNTERS[@]}"; do
        N=$(echo $C | tr ' ' '_' | cut -c1-40)
        rocprofv3 --pmc $C -d $O/pmc_$N -o pmc -- python tools/$T.py $A > $O/pmc_$N.log 2>&1
      done ;;
  esac
done
ls gpurun_out/${R}_*/bench.* | head -20
