cd /root/repo
timeout 300 python -m pytest tests/test_sw_hip.py -x -q --timeout 120 2>&1 | tail -15
