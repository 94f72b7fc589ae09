#!/bin/bash
# Developer tool: sample sclk / power while bench.py runs (is the FP64 kernel clock- or power-limited?)
# usage (on the GPU box): tools/clocks.sh [extra env assignments]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( env "$@" python bench.py --steps ${STEPS:-1500} --warmup 3 > gpurun_out/clocks_bench.json 2>&1 ) &
BP=$!
sleep 8
for i in $(seq 1 200); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use" | sed 's/=*//g; s/GPU\[0\]\s*: //' | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
tail -1 gpurun_out/clocks_bench.json | cut -c1-120
