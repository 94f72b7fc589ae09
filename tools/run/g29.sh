cd /root/repo
timeout 600 python -m pytest tests/test_sw_hip.py tests/test_realign_hip.py tests/test_project_hip.py tests/test_calculate_cigar_hip.py tests/test_engine_hip.py -x -q --timeout 300 2>&1 | tail -3
python tools/realign_small.py 1024x128x8 256x128x8 2>&1 | grep -v amdgpu
PHMM_SW_CHUNKS=1 python tools/realign_small.py 1024x128x8 2>&1 | grep -v amdgpu
timeout 200 python tools/soak_project.py 60 81 2>&1 | tail -1
