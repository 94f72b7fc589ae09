cd /root/repo
python -m pytest tests/test_full_size_configs.py -x -q 2>&1 | tail -8
