cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/r02_ragged; mkdir -p gpurun_out/r02_ragged
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_ragged/trace -o trace -- python bench.py --steps 3 --warmup 1 --main-only --workload ragged > gpurun_out/r02_ragged/trace.log 2>&1
PHMM_TRACE=1 python bench.py --steps 1 --warmup 0 --main-only --workload ragged 2>&1 | grep "phmm plan" | head
