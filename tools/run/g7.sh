cd /root/repo
timeout 120 tools/ubench/salu_mask
