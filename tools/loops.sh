#!/bin/bash
# Developer tool: instruction histogram of the big inner loops of one instantiation (after tools/regs.sh)
# usage: tools/loops.sh 16 19
cd /tmp/st
awk "/^_ZN4phmm12phmm_forwardILi${1}ELi${2}EEE/,/\.amdhsa_kernel/" phmm_kernels-hip-amdgcn-amd-amdhsa-gfx950.s > kk.s
for r in $(grep -n "s_cbranch_scc[01] .LBB[0-9]*_[0-9]*" kk.s | cut -d: -f1); do
  lbl=$(sed -n "${r}p" kk.s | awk '{print $2}'); l0=$(grep -n "^${lbl}:" kk.s | cut -d: -f1)
  if [ -n "$l0" ] && [ $l0 -lt $r ] && [ $((r-l0)) -gt 100 ]; then
    echo "loop $lbl lines $l0-$r  VALU=$(sed -n "${l0},${r}p" kk.s | grep -v '^\s*;' | grep -c '^\s*v_')"
    sed -n "${l0},${r}p" kk.s | grep -v "^\s*;" | awk '{print $1}' | sort | uniq -c | sort -rn | head -${3:-14} | tr '\n' ';'; echo
  fi
done
grep -E "private_segment_fixed_size|vgpr_spill" kk.s | head -2 | tr '\n' ' '; echo
