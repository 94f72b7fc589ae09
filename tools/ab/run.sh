#!/bin/bash
# Developer tool: A/B the same workload over several prebuilt libphmm.so variants on ONE box (box-to-box clocks differ)
cd "$(dirname "$0")/../.."
cp lorikeet_amd/libphmm.so /tmp/libphmm_cur.so
for rep in 1 2; do
for v in prev mul24; do
  cp tools/ab/libphmm_$v.so lorikeet_amd/libphmm.so
  echo "== $v"; python tools/shapes.py --chain config2x1024 2>&1 | grep -v "^phmm plan" | tail -2
done
done
cp /tmp/libphmm_cur.so lorikeet_amd/libphmm.so
