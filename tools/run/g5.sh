cd /root/repo
timeout 300 python -m pytest tests/test_sw_hip.py -x -q --timeout 120 2>&1 | tail -2
for L in 16 8; do echo "L=$L: "; PHMM_SW_LANES=$L python tools/sw_bench.py 1024 2>&1 | tail -2 | cut -c1-100; done
timeout 100 python tools/soak_sw.py 30 13 2>&1 | tail -1
