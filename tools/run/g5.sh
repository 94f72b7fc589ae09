cd /root/repo
timeout 600 python -m pytest tests/test_calculate_cigar_hip.py -x -q --timeout 300 2>&1 | tail -12
