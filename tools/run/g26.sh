cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/eng1; mkdir -p gpurun_out/eng1
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/eng1 -o e --output-format csv -- python tools/engine_call.py 1 > /dev/null 2>&1
python tools/trace_overlap.py gpurun_out/eng1/e_kernel_trace.csv | head -12
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/eng1/e_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=rows[-8:]
t0=int(last[0]["Start_Timestamp"])
for r in last: print("%-50s start %7.1f us dur %6.1f us" % (r["Kernel_Name"][:50], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
rm -rf gpurun_out/eng1
