cd /root/repo
python -m pytest tests/test_cpp_host_layer.py -m gpu -x -q 2>&1 | tail -12
