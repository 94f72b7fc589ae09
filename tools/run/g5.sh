cd /root/repo
timeout 300 python -m pytest tests/test_sw_hip.py -x -q --timeout 100 2>&1 | tail -15
for w in 8 16 24; do PHMM_SW_WAVES_PER_CU=$w python tools/sw_bench.py 1024 2>&1 | tail -1; done
