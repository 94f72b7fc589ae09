cd /root/repo
timeout 400 python tools/soak.py 240 31 2>&1 | tail -1
timeout 300 python tools/soak_engine.py 120 32 2>&1 | tail -1
timeout 300 python tools/soak_sw.py 180 33 2>&1 | tail -1
timeout 300 python tools/soak_project.py 120 34 2>&1 | tail -1
