cd /root/repo
python tools/engine_call.py 1 8 256 1024 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_engine_hip.py tests/test_submit_wait.py -x -q --timeout 600 2>&1 | tail -3
timeout 300 python tools/soak_engine.py 60 52 2>&1 | tail -1
