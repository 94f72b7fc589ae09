#!/bin/bash
# Developer tool: tools/ab/host_ab.py over prebuilt variants, three rounds, on ONE box.
cd "$(dirname "$0")/../.."
cp lorikeet_amd/libphmm.so /tmp/libphmm_cur.so
for rep in 1 2 3; do
for v in ${1:-prev new}; do
  cp tools/ab/libphmm_$v.so lorikeet_amd/libphmm.so
  echo "== $v: $(python tools/ab/host_ab.py 2>&1 | grep -v amdgpu.ids)"
done
done
cp /tmp/libphmm_cur.so lorikeet_amd/libphmm.so
