cd /root/repo
timeout 600 python tools/soak.py 240 7 2>&1 | tail -5
timeout 400 python tools/soak_engine.py 120 3 2>&1 | tail -3
