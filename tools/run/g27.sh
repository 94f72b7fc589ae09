cd /root/repo
timeout 500 python tools/soak.py 300 61 2>&1 | tail -1
timeout 400 python tools/soak_engine.py 240 62 2>&1 | tail -1
timeout 400 python tools/soak_sw.py 300 63 2>&1 | tail -1
timeout 400 python tools/soak_project.py 240 64 2>&1 | tail -1
