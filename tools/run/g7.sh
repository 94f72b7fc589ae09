cd /root/repo
timeout 400 python tools/soak.py 200 91 2>&1 | tail -1
timeout 300 python tools/soak_engine.py 100 92 2>&1 | tail -1
timeout 300 python tools/soak_sw.py 150 93 2>&1 | tail -1
timeout 300 python tools/soak_project.py 120 94 2>&1 | tail -1
