cd /root/repo
cp lorikeet_amd/libphmm.so /tmp/cur.so
for rep in 1 2; do for w in k4_12 k4_19; do cp tools/ab/libphmm_$w.so lorikeet_amd/libphmm.so; echo -n "$w: "; python tools/sw_bench.py 1024 2>&1 | tail -1 | cut -c1-90; done; done
cp /tmp/cur.so lorikeet_amd/libphmm.so
timeout 300 python -m pytest tests/test_sw_hip.py -x -q --timeout 120 2>&1 | tail -2
