cd /root/repo
python tools/sw_bench.py 256 0 haps 2>&1 | grep -v "^{" | tail -2 | cut -c1-200
python tools/sw_bench.py 1024 0 haps 2>&1 | grep -v "^{" | tail -2 | cut -c1-200
PHMM_SW_LANES=8 python tools/sw_bench.py 1024 0 haps 2>&1 | grep -v "^{" | tail -1 | cut -c1-200
