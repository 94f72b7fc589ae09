cd /root/repo
python tools/realign_small.py 2>&1 | grep -v amdgpu
for L in 32 64; do PHMM_SW_LANES=$L timeout 600 python -m pytest tests/test_sw_hip.py tests/test_realign_hip.py -x -q --timeout 200 2>&1 | tail -3; done
timeout 600 python -m pytest tests/test_sw_hip.py tests/test_realign_hip.py -x -q --timeout 200 2>&1 | tail -3
