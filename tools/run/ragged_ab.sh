#!/bin/bash
# gpurun recipe (A/B): the mixed batch resident and through host buffers -- the per-read classes in front of the forked chained
# launches (default) or behind the join (PHMM_CLASSES_LAST=1), three times each on one box; then the mixed-batch parity tests
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2 3; do
for v in "" "PHMM_CLASSES_LAST=1"; do
  echo -n "[$v] "
  env $v BENCH_ROWS=ragged python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])['rows']['ragged']; print('resident %.1f GCUPS %.3f ms; host %.1f GCUPS %.2f ms' % (r['gcups'], r['ms'], r['host_gcups'], r['host_ms']))"
done; done
python -m pytest tests/test_full_size_configs.py tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_f32_first.py -m gpu -q -x 2>&1 | tail -3
} > gpurun_out/r05_ragged_ab.txt 2>&1
cat gpurun_out/r05_ragged_ab.txt
