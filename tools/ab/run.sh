#!/bin/bash
# Developer tool: A/B the same workloads over several prebuilt libphmm.so variants on ONE box (box-to-box clocks differ)
# usage (through gpurun): tools/ab/run.sh "<variants>" "<shapes.py arguments>"   with tools/ab/libphmm_<variant>.so present
cd "$(dirname "$0")/../.."
cp lorikeet_amd/libphmm.so /tmp/libphmm_cur.so
for rep in 1 2; do
for v in ${1:-prev new}; do
  cp tools/ab/libphmm_$v.so lorikeet_amd/libphmm.so
  echo "== $v"; python tools/shapes.py ${2:---chain config2x1024} 2>&1 | grep -v "^phmm plan\|amdgpu.ids" | grep "chain=16\|L=0"
done
done
cp /tmp/libphmm_cur.so lorikeet_amd/libphmm.so
