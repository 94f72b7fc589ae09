"""Timeline of one phmm_region_compute call from a rocprofv3 kernel trace (csv): start / end of every kernel of a call in
the middle of the run, relative to the call's first kernel.  usage: python tools/region_timeline.py <dir with *_kernel_trace.csv> [call index]"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
# a call starts at a phmm_prep_reads kernel (or the aligner's, whichever comes first after a gap)
starts = [i for i, k in enumerate(ks) if "phmm_prep_reads" in k[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
i0 = starts[n]
while i0 > 0 and ks[i0][0] - ks[i0 - 1][1] < 3000 and not any(x in ks[i0 - 1][2] for x in ("phmm_pick", "phmm_project")):
    i0 -= 1
i1 = starts[n + 1] if n + 1 < len(starts) else len(ks)
t0 = ks[i0][0]
print("call %d of %d (us from the first kernel's start; queue)" % (n, len(starts)))
for s, e, name, q in ks[i0:i1 + 1]:
    if s - t0 > 400000:
        break
    print("%8.1f %8.1f  %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name[:90]))
