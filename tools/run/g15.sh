cd /root/repo
for kb in 64 256 1024; do echo "zero-copy limit $kb KB"; PHMM_ZERO_COPY_OUT_KB=$kb python tools/hostpath_small.py 8 9 12 16 24 2>&1 | grep regions | grep -v plan; done
