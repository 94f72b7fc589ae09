cd /root/repo
timeout 500 python tools/soak_sw.py 180 5 2>&1 | tail -3
