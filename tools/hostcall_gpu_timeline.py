"""GPU-side picture of ONE host-buffer call from a rocprofv3 kernel trace + memory-copy trace (csv): when each chunk's kernels
ran, how long the device computed, how long it sat idle inside the call.  usage: python tools/hostcall_gpu_timeline.py <dir>"""
import csv
import glob
import sys

d = sys.argv[1]
ks = []
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
ks.sort()
# the last call: kernels after the last gap of > 3 ms
cut = 0
for i in range(1, len(ks)):
    if ks[i][0] - max(k[1] for k in ks[max(0, i - 50):i]) > 3_000_000:
        cut = i
call = ks[cut:]
t0, t1 = call[0][0], max(k[1] for k in call)
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in call:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("last call: %d kernels, span %.2f ms, device busy (union) %.2f ms, idle inside the span %.2f ms, sum of kernel times %.2f ms"
      % (len(call), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, sum(e - s for s, e, _, _ in call) / 1e6))
for s, e, n, q in call:
    if e - s > 200_000:
        print("%8.2f %8.2f %7.2f q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
