#!/bin/bash
# Developer tool: launch-order cost models of a multi-stream item (an experimental build that reads PHMM_ORDER_MODE, variant exp2:
# 0 the whole run's rows, as shipped; 1 its longest stream; 2 / 3 / 4 the longest stream x 1.5 / 2 / 3; 5 the run's rows x 2;
# 6 every multi-stream item first), tools/ab/exp.py each, on ONE box.
cd "$(dirname "$0")/../.."
cp lorikeet_amd/libphmm.so /tmp/libphmm_cur.so
cp tools/ab/libphmm_exp2.so lorikeet_amd/libphmm.so
for rep in 1 2; do
for m in 0 1 2 3 4 5 6; do
  echo "order mode $m : $(PHMM_ORDER_MODE=$m python tools/ab/exp.py 2>&1 | grep -v amdgpu.ids)"
done
done
cp /tmp/libphmm_cur.so lorikeet_amd/libphmm.so
