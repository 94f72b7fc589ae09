cd /root/repo
for L in 0 64 32 16; do echo "== PHMM_SW_LANES=$L"; PHMM_SW_LANES=$L python tools/realign_small.py 16x128x8 32x128x8 64x128x8 128x128x8 2>&1 | grep -v amdgpu; done
