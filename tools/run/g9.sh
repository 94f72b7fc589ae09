cd /root/repo
for w in 16 20 24 28 32; do echo "waves/CU $w"; PHMM_SW_WAVES_PER_CU=$w python tools/sw_bench.py 1024 2>&1 | tail -1 | cut -c1-120; done
