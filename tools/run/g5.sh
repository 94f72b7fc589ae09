cd /root/repo
timeout 300 python -m pytest tests/test_sw_hip.py -x -q --timeout 120 2>&1 | tail -5
python tools/sw_bench.py 1024 2>&1 | tail -1 | cut -c1-130
