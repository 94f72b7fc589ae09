#!/bin/bash
# Developer tool: how a caller waits for the region server (an experimental build that reads PHMM_SERVER_WAIT, variant exp3:
# 0 a short spin then 20 us naps, as shipped; 1 spinning while the callers fit the cores; 2 / 3 naps until 70 / 85 % of the recent
# time of a call, then spinning), private handles, 8 / 10 / 16 / 32 threads, three shapes, on ONE box.
cd "$(dirname "$0")/../.."
cp lorikeet_amd/libphmm.so /tmp/libphmm_cur.so
cp tools/ab/libphmm_exp3.so lorikeet_amd/libphmm.so
for rep in 1 2; do
for m in 0 1 2 3; do
  echo "== wait mode $m"
  PHMM_SERVER_WAIT=$m TB_MODE=fused TB_THREADS=8,10,16,32 tools/threads_bench 1 | grep fused
  PHMM_SERVER_WAIT=$m TB_MODE=fused TB_THREADS=10,16 tools/threads_bench 1 30 3 100 200 | grep fused | sed 's/^/30x3  /'
  PHMM_SERVER_WAIT=$m TB_SHAPE=ragged TB_MODE=fused TB_THREADS=10,16 tools/threads_bench 1 | grep fused | sed 's/^/ragged /'
done
done
cp /tmp/libphmm_cur.so lorikeet_amd/libphmm.so
