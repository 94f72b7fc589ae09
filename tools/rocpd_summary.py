"""Developer tool: turn rocprofv3 rocpd .db outputs (gpurun_out/<tag>/...) into the text/JSON summaries
committed under profiles/.   usage: python tools/rocpd_summary.py gpurun_out/r01 profiles/r01"""
import glob, json, os, sqlite3, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
lines = []
tr = glob.glob(os.path.join(src, "trace", "*.db"))
if tr:
    c = sqlite3.connect(tr[0]).cursor()
    lines.append("== rocprofv3 --kernel-trace --stats : kernel summary (durations in ns) ==")
    lines.append("%-64s %8s %16s %14s %8s" % ("name", "calls", "total_ns", "avg_ns", "pct"))
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append("%-64s %8d %16.0f %14.0f %8.3f" % (r[0][:64], r[1], r[2] * 1e3, r[3] * 1e3, r[4]))
    # steady state: the TIMED launches of bench.py are the LAST `steps` dispatches of every kernel of a launch (the warm-up
    # launches come first), so their mean is the figure the bench line reports -- no warm-up dispatch in it
    bj0 = os.path.join(src, "bench.json")
    steps = json.loads(open(bj0).read().strip().splitlines()[-1]).get("steps") if os.path.exists(bj0) else None
    if steps:
        lines.append("")
        lines.append("== steady state: mean over the last %d launches' dispatches of each kernel (the timed loop; warm-up excluded) ==" % steps)
        lines.append("%-64s %8s %14s %14s %14s" % ("name", "calls", "avg_ns_all", "avg_ns_timed", "per_launch"))
        names = [r[0] for r in c.execute("select name from top_kernels where name like '%phmm%'")]
        for n in names:
            d = [r[0] for r in c.execute("select duration from kernels where name = ? order by start", (n,))]
            # a launch may dispatch a kernel several times (K ranges, rescue): dispatches per launch = total / (steps + warm-up)
            total_launches = steps + json.loads(open(bj0).read().strip().splitlines()[-1]).get("warmup", 0)
            per = max(1, len(d) // max(1, total_launches))
            timed = d[-steps * per:]
            lines.append("%-64s %8d %14.0f %14.0f %14d" % (n[:64], len(d), sum(d) / len(d), sum(timed) / len(timed), per))
    lines.append("")
    lines.append("== per-dispatch (first 12) ==")
    for r in c.execute("select name, duration, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size from kernels where name like '%phmm%' limit 12"):
        lines.append("%s dur_ns=%d grid=(%d,%d) wg=%d lds=%d vgpr=%d agpr=%d sgpr=%d scratch=%d" % r)
pmc = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*.db"))):
    c = sqlite3.connect(f).cursor()
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%phmm%' group by kernel_name, counter_name"):
        pmc.setdefault(r[0], {})[r[1]] = {"avg_per_dispatch": r[2], "dispatches": r[3], "avg_duration_ns": r[4]}
if pmc:
    lines.append("")
    lines.append("== rocprofv3 --pmc passes (separate runs, averages per dispatch) ==")
    for k, v in pmc.items():
        lines.append(k)
        for n, e in sorted(v.items()):
            lines.append("   %-24s %16.6g   (n=%d, avg dispatch %.0f ns)" % (n, e["avg_per_dispatch"], e["dispatches"], e["avg_duration_ns"]))
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    lines.append("")
    lines.append("== bench.py line of the same command ==")
    lines.append(open(bj).read().strip().splitlines()[-1])
open(dst + "_summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(pmc, open(dst + "_pmc.json", "w"), indent=1)
print("\n".join(lines))
