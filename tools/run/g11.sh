cd /root/repo
export TMPDIR=/tmp
bash tools/profile.sh r02_config2_f64 --workload config2 > gpurun_out/prof_a.log 2>&1
bash tools/profile.sh r02_config2_f32 --workload config2 --f32-first > gpurun_out/prof_b.log 2>&1
bash tools/profile.sh r02_config5 --workload config5 > gpurun_out/prof_c.log 2>&1
bash tools/profile.sh r02_config3 --workload config3 > gpurun_out/prof_d.log 2>&1
bash tools/profile.sh r02_ragged --workload ragged > gpurun_out/prof_e.log 2>&1
# engine call and Smith-Waterman: kernel trace + instruction counters of the tools that drive them
for T in engine_call sw_bench; do
  O=gpurun_out/r02_$T; rm -rf $O; mkdir -p $O
  A=""; [ $T = engine_call ] && A="256"; [ $T = sw_bench ] && A="1024"
  python tools/$T.py $A > $O/bench.txt 2>&1
  rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python tools/$T.py $A > $O/trace.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C -d $O/pmc_$N -o pmc -- python tools/$T.py $A > $O/pmc_$N.log 2>&1
  done
done
ls gpurun_out/r02_*/bench.* | head -20
