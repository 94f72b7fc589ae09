cd /root/repo
timeout 400 python tools/soak.py 240 41 2>&1 | tail -1
timeout 300 python tools/soak_engine.py 120 42 2>&1 | tail -1
timeout 300 python tools/soak_sw.py 180 43 2>&1 | tail -1
timeout 300 python tools/soak_project.py 120 44 2>&1 | tail -1
