#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rows in calls,sw calls,ragged,sw; do
  echo "== BENCH_ROWS=$rows"
  BENCH_ROWS=$rows python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rows']
print('sw', r.get('smith_waterman',{}).get('ms'), r.get('smith_waterman',{}).get('kernel_ms'), 'ragged', (r.get('ragged') or {}).get('host_ms'))"
done
} > gpurun_out/r05_sw_calls3.txt 2>&1
cat gpurun_out/r05_sw_calls3.txt
