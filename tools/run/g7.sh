cd /root/repo
timeout 200 python tools/soak.py 90 21 2>&1 | tail -2
timeout 200 python tools/soak_engine.py 60 22 2>&1 | tail -2
timeout 200 python tools/soak_sw.py 60 23 2>&1 | tail -1
