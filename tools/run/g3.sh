cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/r02_ragged_t; mkdir -p gpurun_out/r02_ragged_t
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_ragged_t/trace -o trace -- python bench.py --steps 3 --warmup 1 --main-only --workload ragged > gpurun_out/r02_ragged_t/trace.log 2>&1
