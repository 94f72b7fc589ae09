#!/usr/bin/env python3
"""How many kernels of a rocprofv3 --kernel-trace CSV run at the same time, and on which queues.

    python tools/trace_overlap.py <kernel_trace.csv> [<memory_copy_trace.csv>]

Prints the time-weighted distribution of the number of kernels in flight, the busy fraction and dispatch count of every
queue, and the same for the copies.  (The per-region pipeline of tools/threads_bench: are concurrent callers' kernels
overlapping on the device, or queueing behind each other?)"""
import csv
import os
import sys
from collections import Counter, defaultdict


def load(path):
    rows = list(csv.DictReader(open(path)))
    out = []
    for r in rows:
        out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Direction", "")), r.get("Kernel_Name", r.get("Direction", ""))))
    return out


def concurrency(iv):
    ev = []
    for s, e, _, _ in iv:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist = Counter()
    level, last = 0, ev[0][0]
    for t, d in ev:
        hist[level] += t - last
        last = t
        level += d
    total = sum(hist.values())
    return {k: v / total for k, v in sorted(hist.items())}, total


def main():
    iv = load(sys.argv[1])
    tail = float(os.environ.get("TRACE_TAIL", "1"))  # e.g. 0.4: only the last 40 % of the run (warm-up and set-up excluded)
    if tail < 1:
        t0, t1 = min(s for s, _, _, _ in iv), max(e for _, e, _, _ in iv)
        iv = [x for x in iv if x[0] >= t1 - tail * (t1 - t0)]
    hist, total = concurrency(iv)
    print("%d dispatches over %.1f ms" % (len(iv), total / 1e6))
    print("kernels in flight (share of the time): " + ", ".join("%d: %.1f %%" % (k, 100 * v) for k, v in hist.items()))
    print("mean in flight %.2f" % sum(k * v for k, v in hist.items()))
    byk = defaultdict(list)
    for s, e, _, k in iv:
        byk[k.split("(")[0][-40:]].append(e - s)
    for k, d in sorted(byk.items(), key=lambda kv: -sum(kv[1])):
        d.sort()
        print("  %-42s %6d x  mean %7.1f us  median %7.1f  p90 %7.1f" % (k, len(d), sum(d) / len(d) / 1e3, d[len(d) // 2] / 1e3, d[len(d) * 9 // 10] / 1e3))
    byq = defaultdict(list)
    for s, e, q, _ in iv:
        byq[q].append((s, e))
    for q, lst in sorted(byq.items()):
        busy = sum(e - s for s, e in lst)
        lst.sort()
        overl = sum(1 for (a, b), (c, d) in zip(lst, lst[1:]) if c < b)
        print("  queue %s: %d dispatches, busy %.1f %%, %d overlapping their predecessor" % (q, len(lst), 100.0 * busy / total, overl))
    if len(sys.argv) > 2:
        cp = load(sys.argv[2])
        h2, t2 = concurrency(cp)
        print("%d copies; in flight: " % len(cp) + ", ".join("%d: %.1f %%" % (k, 100 * v) for k, v in h2.items()))


if __name__ == "__main__":
    main()
