#!/bin/bash
cd "$(dirname "$0")/../.."
for i in 1 2 3; do
  echo "== run $i"; PHMM_MIRROR_CANARY=1 TB_VERIFY=1 TB_SHAPE=ragged TB_MODE=own TB_THREADS=1,2,3,4,8,16 timeout 300 tools/threads_bench 1.0 2>&1 | tail -8
done
for i in 1 2; do
echo "== fused run $i"; PHMM_MIRROR_CANARY=1 TB_VERIFY=1 TB_SHAPE=ragged TB_MODE=fused TB_THREADS=1,2,3,4,8,16 timeout 300 tools/threads_bench 1.0 2>&1 | tail -8
done
