cd /root/repo
timeout 600 python -m pytest tests/test_sw_hip.py tests/test_realign_hip.py tests/test_project_hip.py tests/test_calculate_cigar_hip.py -x -q --timeout 300 2>&1 | tail -3
TB_MODE=pipeline TB_THREADS=1,4,8,16 tools/threads_bench 1.5
python tools/realign_small.py 2>&1 | grep -v amdgpu | head -3
echo "--- copies instead of zero-copy ---"
PHMM_SW_NO_ZERO_COPY=1 TB_MODE=pipeline TB_THREADS=1,4,8 tools/threads_bench 1.5
PHMM_SW_NO_ZERO_COPY=1 python tools/realign_small.py 2>&1 | grep -v amdgpu | head -3
