cd /root/repo
timeout 500 python tools/soak.py 240 71 2>&1 | tail -1
timeout 400 python tools/soak_engine.py 180 72 2>&1 | tail -1
timeout 400 python tools/soak_sw.py 300 73 2>&1 | tail -1
timeout 400 python tools/soak_project.py 180 74 2>&1 | tail -1
