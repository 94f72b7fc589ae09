#!/bin/bash
# Developer tool: the naps of the region server's waiters with the thread's timer slack tightened (an experimental build that reads
# PHMM_SLACK_NS, variant exp4: 0 = the default 50 us slack), private handles, 16 / 24 / 32 threads, on ONE box.
cd "$(dirname "$0")/../.."
cp lorikeet_amd/libphmm.so /tmp/libphmm_cur.so
cp tools/ab/libphmm_exp4.so lorikeet_amd/libphmm.so
for rep in 1 2; do
for ns in 0 1000 10000; do
  echo "== slack $ns ns"
  PHMM_SLACK_NS=$ns TB_MODE=fused TB_THREADS=16,24,32 tools/threads_bench 1 | grep fused
  PHMM_SLACK_NS=$ns TB_SHAPE=ragged TB_MODE=fused TB_THREADS=16,32 tools/threads_bench 1 | grep fused | sed 's/^/ragged /'
done
done
cp /tmp/libphmm_cur.so lorikeet_amd/libphmm.so
