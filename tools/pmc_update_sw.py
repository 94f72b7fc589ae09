"""Developer tool: fold the PMC passes of `tools/sw_bench.py` (tools/run/profiles.sh) into profiles/pmc_traffic.json as the
entry bench.py's smith_waterman row reads (workload "smith_waterman", keyed by the alignments of the call and the hash of
the kernel sources, like the PairHMM entries of tools/pmc_update.py).  Counters are summed over the kernels of one call
(a large call is several pieces).    usage: python tools/pmc_update_sw.py gpurun_out/<tag> profiles/<name>"""
import json
import os
import sys

src, name = sys.argv[1], sys.argv[2]
info = None
for line in open(os.path.join(src, "bench.txt")):
    if line.startswith("{") and "sw_bench" in line:
        info = json.loads(line)["sw_bench"]
pmc = json.load(open(name + "_pmc.json"))
kern = max(pmc, key=lambda k: max(v["avg_duration_ns"] * v["dispatches"] for v in pmc[k].values()))
c = {cn: v["avg_per_dispatch"] * v["dispatches"] / info["calls"] for cn, v in pmc[kern].items()}
entry = {
    "workload": "smith_waterman", "regions": info["alignments"], "precision": "i32", "kernel": kern,
    "kernel_short": "phmm_sw_align_kernel", "src_hash": info["src_hash"],
    "fetch_size_kb": c.get("FETCH_SIZE"), "write_size_kb": c.get("WRITE_SIZE"), "fetch_size_correction": 2.0,
    "hbm_bytes_per_launch": int(2.0 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024),
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c else None,
    "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
    "lds_insts_per_launch": c.get("SQ_INSTS_LDS"), "wait_inst_any": c.get("SQ_WAIT_INST_ANY"),
    "wave_cycles": c.get("SQ_WAVE_CYCLES"), "waves": c.get("SQ_WAVES"),
    "kernel_ms_bench": info["kernel_ms"], "cells_per_launch": info["cells"], "backtrack_flag_bytes_per_launch": info["backtrack_bytes"],
    "clock_mhz": info["clock_mhz"], "source": os.path.basename(name) + "_summary.txt",
    "note": "one `launch` = one phmm_sw_align call (its pieces summed); rocprofv3 --pmc passes of tools/sw_bench.py; "
            "reads = 2 x FETCH_SIZE, writes = WRITE_SIZE as for the PairHMM entries",
}
path = os.path.join(os.path.dirname(name) or ".", "pmc_traffic.json")
entries = json.load(open(path))
key = lambda e: (e.get("workload"), e.get("regions"), e.get("precision", "f64"))  # noqa: E731
entries = [e for e in entries if key(e) != key(entry)] + [entry]
json.dump(entries, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
