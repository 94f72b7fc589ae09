"""Developer tool: fold the PMC passes of one tools/profile.sh run into profiles/pmc_traffic.json, the file bench.py
reads for roofline.traffic / l2_hit_rate / valu_issue.  An entry is keyed by (workload, regions, precision, dominant
kernel, hash of lorikeet_amd/csrc/*.hip,*.hpp) -- all taken from the bench line of that very run -- so bench.py can
tell a measurement of the kernels it is running from a stale one.

usage: python tools/pmc_update.py gpurun_out/<tag> profiles/<name>        (after tools/rocpd_summary.py made <name>_pmc.json)
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per read request although every request
on gfx950 is a 128-B line fill (calibrated for this access pattern in round 1, see the note of the entry), so
reads = 2 x FETCH_SIZE KB; WRITE_SIZE as counted."""
import json
import os
import sys

src, name = sys.argv[1], sys.argv[2]
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
pmc = json.load(open(name + "_pmc.json"))
# dominant kernel of the run = the PMC row with the largest total time; one bench launch (phmm_batch_launch) may consist of
# several kernels (a mixed batch: one per lanes-per-pair value, the f64 redo of the f32-first mode, the exact pass), so
# every counter is summed over all of them and divided by the number of launches
kern = max(pmc, key=lambda k: max(v["avg_duration_ns"] * v["dispatches"] for v in pmc[k].values()))
launches = max(v["dispatches"] for v in pmc[kern].values())
c = {}
for k in pmc:
    for cname, v in pmc[k].items():
        c[cname] = c.get(cname, 0.0) + v["avg_per_dispatch"] * v["dispatches"] / launches
entry = {
    "workload": bench["config"]["workload"].split(" ")[0], "regions": bench["config"]["regions_per_gpu"],
    "precision": "f32_first" if bench["dtype"].startswith("f32") else "f64",
    "kernel": kern, "kernels_summed": sorted(pmc), "kernel_short": bench["roofline"]["kernel"], "src_hash": bench["roofline"]["src_hash"],
    "fetch_size_kb": c.get("FETCH_SIZE"), "write_size_kb": c.get("WRITE_SIZE"), "fetch_size_correction": 2.0,
    "hbm_bytes_per_launch": int(2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024),
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c else None,
    "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
    "lds_insts_per_launch": c.get("SQ_INSTS_LDS"), "wait_inst_any": c.get("SQ_WAIT_INST_ANY"),
    "wave_cycles": c.get("SQ_WAVE_CYCLES"), "busy_cycles": c.get("SQ_BUSY_CYCLES"), "waves": c.get("SQ_WAVES"),
    "lds_bank_conflict": c.get("SQ_LDS_BANK_CONFLICT"),
    "kernel_ms_bench": bench["roofline"]["kernel_ms"], "cells_per_launch": bench["config"]["cells_per_gpu_per_step"],
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "source": os.path.basename(name) + "_summary.txt",
    "note": "rocprofv3 --pmc passes of `bench.py --main-only` (tools/profile.sh), averages per dispatch of the dominant "
            "kernel; reads = 2 x FETCH_SIZE (gfx950: 64 B counted per 128-B line fill; calibrated with --flush-caches in "
            "round 1: FETCH_SIZE unchanged with cold caches, EA read requests == L2 misses), writes = WRITE_SIZE",
}
path = os.path.join(os.path.dirname(name) or ".", "pmc_traffic.json")
try:
    entries = json.load(open(path))
except Exception:
    entries = []
key = lambda e: (e.get("workload"), e.get("regions"), e.get("precision", "f64"))  # noqa: E731
entries = [e for e in entries if key(e) != key(entry)] + [entry]
json.dump(entries, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
