cd /root/repo
for kb in 4096 8192 16384; do echo "PHMM_CHUNK_KB=$kb"; PHMM_CHUNK_KB=$kb python tools/hostpath_ragged.py 2>&1 | grep "call\|chunks" | tail -3; done
