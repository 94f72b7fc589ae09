cd /root/repo
python tools/engine_call.py 1 8 2>&1 | grep -v amdgpu
PHMM_STAGE_IN_KB=0 python tools/engine_call.py 1 8 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_engine_hip.py tests/test_engine_submit.py tests/test_hip_parity.py -x -q --timeout 600 2>&1 | tail -3
