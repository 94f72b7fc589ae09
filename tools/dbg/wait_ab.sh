#!/bin/bash
cd "$(dirname "$0")/../.."
for cfg in "512 0" "64 0" "64 1" "2000000000 0"; do
  set -- $cfg
  echo "== spins $1 mode $2"
  PHMM_SERVER_WAIT_SPINS=$1 PHMM_SERVER_WAIT_MODE=$2 TB_MODE=fused TB_THREADS=10,16,32 tools/threads_bench 1 | grep fused
done
echo "== ragged"
TB_SHAPE=ragged TB_MODE=fused TB_THREADS=10,16 tools/threads_bench 1 | grep fused
nproc; python3 -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"
