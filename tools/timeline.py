"""Developer tool: merge rocprofv3 --kernel-trace --memory-copy-trace CSVs into one GPU timeline (last N events).
usage: python tools/timeline.py <rocprof output dir> [N]"""
import csv, glob, sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
ev = []
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel " + r["Kernel_Name"].split("(")[0][-28:]))
for r in csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy   " + r["Direction"].replace("MEMORY_COPY_", "")))
ev.sort()
sel = ev[-n:]
t0 = sel[0][0]
print("%10s %10s %10s  %s" % ("start us", "end us", "dur us", "what"))
for s, e, what in sel:
    print("%10.1f %10.1f %10.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, what))
