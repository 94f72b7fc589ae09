cd /root/repo
bash tools/profile.sh r02_config2_f64 --workload config2 > gpurun_out/prof_f64.log 2>&1
bash tools/profile.sh r02_config2_f32 --workload config2 --f32-first > gpurun_out/prof_f32.log 2>&1
tail -3 gpurun_out/prof_f64.log gpurun_out/prof_f32.log
