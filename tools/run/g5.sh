cd /root/repo
python tools/hostpath_ragged.py 2>&1 | grep "call\|chunks\|enqueue" | head -14
