cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r02_threads_bench.txt
{
echo "# tools/threads_bench on MI355X: one region per call from T C++ host threads, host buffers (PCIe included)"
echo "## PairHMM alone (phmm_compute), private handles and one shared handle"
TB_THREADS=1,2,4,8,16,32 tools/threads_bench 1
echo "## PairHMM alone, small regions (30 reads x 3 haplotypes)"
TB_MODE=own TB_THREADS=1,8,16 tools/threads_bench 1 30 3 100 120
echo "## realignment alone (phmm_realign_reads on likelihoods computed beforehand)"
TB_MODE=realign TB_THREADS=1,2,4,8,16 tools/threads_bench 1
echo "## both per region (phmm_compute, then phmm_realign_reads with its likelihoods)"
TB_MODE=pipeline TB_THREADS=1,2,4,8,16 tools/threads_bench 1
echo "## both per region, 8 regions per call"
TB_MODE=pipeline TB_THREADS=1,2,4,8 tools/threads_bench 1 128 8 150 300 8
echo "## both per region, 16 hardware queues (GPU_MAX_HW_QUEUES=16)"
GPU_MAX_HW_QUEUES=16 TB_MODE=pipeline TB_THREADS=8,16 tools/threads_bench 1
echo "## the same without the small-call shortcuts (results by copies, inputs by the copy engine)"
PHMM_SW_NO_ZERO_COPY=1 PHMM_STAGE_IN_KB=0 TB_MODE=pipeline TB_THREADS=1,8 tools/threads_bench 1
for q in 4 16; do
echo "## kernels in flight, 8 threads, both per region, $q hardware queues (rocprofv3 --kernel-trace, tools/trace_overlap.py)"
rm -rf gpurun_out/tb8; mkdir -p gpurun_out/tb8
GPU_MAX_HW_QUEUES=$q TB_MODE=pipeline TB_THREADS=8 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tb8 -o tb8 --output-format csv -- tools/threads_bench 0.4 2>&1 | grep threads:
python tools/trace_overlap.py gpurun_out/tb8/tb8_kernel_trace.csv gpurun_out/tb8/tb8_memory_copy_trace.csv
done
rm -rf gpurun_out/tb8
} > $O 2>&1
tail -5 $O
