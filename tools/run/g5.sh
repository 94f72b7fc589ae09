cd /root/repo
timeout 600 python -m pytest tests/test_realign_hip.py -x -q --timeout 200 2>&1 | tail -12
