cd /root/repo
timeout 600 python -m pytest tests/test_sw_hip.py tests/test_realign_hip.py tests/test_project_hip.py tests/test_calculate_cigar_hip.py -x -q --timeout 300 2>&1 | tail -3
python tools/sw_bench.py 1024 2>&1 | tail -3
python tools/sw_bench.py 1024 0 haps 2>&1 | tail -2
python tools/realign_small.py 2>&1 | grep -v amdgpu | head -3
timeout 300 python tools/soak_sw.py 45 11 2>&1 | tail -1
