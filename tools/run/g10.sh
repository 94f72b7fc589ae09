cd /root/repo
PHMM_TRACE=1 python bench.py --steps 2 --warmup 1 --main-only --workload ragged 2>&1 | grep "class\|phmm plan" | sort -k2 | head -90
