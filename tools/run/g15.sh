cd /root/repo
TRACE_N=1 python tools/hostpath_small.py 1 2>&1 | grep -v "class" | tail -8
