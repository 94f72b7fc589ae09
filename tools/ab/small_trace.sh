#!/bin/bash
# Developer tool: per-kernel durations of a 200-region ragged launch (launches one at a time) over prebuilt variants, on ONE box.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
cp lorikeet_amd/libphmm.so /tmp/libphmm_cur.so
for v in ${1:-prev new}; do
  cp tools/ab/libphmm_$v.so lorikeet_amd/libphmm.so
  rm -rf /tmp/st_$v; rocprofv3 --kernel-trace --stats -d /tmp/st_$v -o t --output-format csv -- python tools/ab/small_resident.py ${2:-200} > /tmp/st_$v.log 2>&1
  echo "== $v"; python - "$(find /tmp/st_$v -name "*kernel_stats.csv" | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-66s calls %4s  avg %10.1f us  min %10.1f  max %10.1f" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
cp /tmp/libphmm_cur.so lorikeet_amd/libphmm.so
