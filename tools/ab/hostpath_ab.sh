#!/bin/bash
# Developer tool: tools/hostpath_ragged.py (per-chunk trace of one phmm_compute call over the ragged mix) over prebuilt variants, on ONE box.
cd "$(dirname "$0")/../.."
cp lorikeet_amd/libphmm.so /tmp/libphmm_cur.so
for v in ${1:-prev new}; do
  cp tools/ab/libphmm_$v.so lorikeet_amd/libphmm.so
  echo "== $v"; python tools/hostpath_ragged.py 2>&1 | grep -v "amdgpu.ids\|^  class\|phmm plan" 
done
cp /tmp/libphmm_cur.so lorikeet_amd/libphmm.so
