cd /root/repo
TB_MODE=own TB_THREADS=1,4,8 tools/threads_bench 1
PHMM_STAGE_IN_KB=0 TB_MODE=own TB_THREADS=1,4,8 tools/threads_bench 1
TB_MODE=own TB_THREADS=1,8,16 tools/threads_bench 1 30 3 100 120
PHMM_STAGE_IN_KB=0 TB_MODE=own TB_THREADS=1,8,16 tools/threads_bench 1 30 3 100 120
TB_MODE=pipeline TB_THREADS=1,8 tools/threads_bench 1
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -3
