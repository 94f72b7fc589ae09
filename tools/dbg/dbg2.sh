#!/bin/bash
cd "$(dirname "$0")/../.."
for mode in default noall; do
  for i in 1 2 3 4 5 6; do
    if [ $mode = noall ]; then export PHMM_REGION_SW_ALL=0; else unset PHMM_REGION_SW_ALL; fi
    echo "== $mode $i"; timeout 60 python tools/dbg/dbg2.py 21 5 2>&1 | grep -v amdgpu.ids | tail -3
  done
done
