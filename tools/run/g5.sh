cd /root/repo
timeout 600 python -m pytest tests/test_project_hip.py -x -q --timeout 300 2>&1 | tail -5
timeout 200 python tools/soak_project.py 60 3 2>&1 | tail -3
