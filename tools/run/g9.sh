cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/soak_sw.py 60 9 2>&1 | tail -1
for T in sw_bench; do
  O=gpurun_out/r02_$T; rm -rf $O; mkdir -p $O
  A="1024"
  python tools/$T.py $A > $O/bench.txt 2>&1
  rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python tools/$T.py $A > $O/trace.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C -d $O/pmc_$N -o pmc -- python tools/$T.py $A > $O/pmc_$N.log 2>&1
  done
done
