#!/bin/bash
cd "$(dirname "$0")/../.."
for f in 0 1 2 3; do
  export PHMM_SERVER_FENCES=$f
  ok=0
  for i in 1 2 3; do
    timeout 120 python -m pytest tests/test_server_hip.py -x -q -m gpu -k "launched_pipeline or oracle_pipeline or beside_other" 2>&1 | tail -1 | grep -q "passed" && ok=$((ok+1))
  done
  echo "fences=$f uncached: $ok/3 runs green"
  timeout 60 tools/server_trace 1 200 | sed -n '1p;5p;6p'
  timeout 60 tools/server_trace 10 200 | sed -n '1p;5p;8p'
done
unset PHMM_SERVER_FENCES
export PHMM_SERVER_CACHED=1
ok=0
for i in 1 2 3; do
  timeout 120 python -m pytest tests/test_server_hip.py -x -q -m gpu -k "launched_pipeline or oracle_pipeline or beside_other" 2>&1 | tail -1 | grep -q "passed" && ok=$((ok+1))
done
echo "cached, fences=3: $ok/3 runs green"
timeout 60 tools/server_trace 10 200 | sed -n '1p;5p;8p'
