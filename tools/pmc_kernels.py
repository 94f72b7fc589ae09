"""Per-kernel means of a `rocprofv3 --pmc ... --output-format csv` run: for every kernel name the number of dispatches and
the mean of every counter per dispatch.  usage: python tools/pmc_kernels.py <dir> [substring of the kernel names to keep]"""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = sys.argv[2] if len(sys.argv) > 2 else ""
for name in sorted(rows):
    if keep not in name:
        continue
    c = rows[name]
    n = max(len(v) for v in c.values())
    print("%s  (%d dispatches)" % (name[:100], n))
    print("   " + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
