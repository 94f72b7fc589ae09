cd /root/repo
export TMPDIR=/tmp
bash tools/profile.sh r02_ragged --workload ragged > gpurun_out/prof_e.log 2>&1
ls gpurun_out/r02_ragged/bench.json
