"""bench.py's bookkeeping that can be checked without a GPU: every kernel file belongs to a hashed family (so a PMC entry of
profiles/pmc_traffic.json cannot outlive a change of the kernels it measured), stale or foreign entries are refused, and
the committed entries carry the keys bench.py reads."""
import glob
import json
import os

import bench
from conftest import ROOT


def test_every_kernel_source_is_in_a_hashed_family():
    hips = {os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "lorikeet_amd", "csrc", "*.hip"))}
    listed = set().union(*bench.KERNEL_SOURCES.values())
    assert hips <= listed, hips - listed
    for fam in bench.KERNEL_SOURCES.values():
        for name in fam:
            assert os.path.exists(os.path.join(ROOT, "lorikeet_amd", "csrc", name)), name
    assert bench.source_hash("pairhmm") != bench.source_hash("sw")


def test_pmc_entries_are_refused_unless_workload_kernel_and_sources_match(tmp_path, monkeypatch):
    entries = [{"workload": "config2", "regions": 1024, "precision": "f64", "kernel_short": "phmm_forward_chain<16,19>",
                "src_hash": bench.source_hash("pairhmm"), "hbm_bytes_per_launch": 1},
               {"workload": "smith_waterman", "regions": 131072, "precision": "i32", "kernel_short": "phmm_sw_align_kernel",
                "src_hash": "0123456789abcdef", "hbm_bytes_per_launch": 2}]
    os.makedirs(tmp_path / "profiles")
    json.dump(entries, open(tmp_path / "profiles" / "pmc_traffic.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_hash", lambda family="pairhmm": entries[0]["src_hash"] if family == "pairhmm" else "feedfeedfeedfeed")
    e, why = bench.pmc_entry("config2", 1024, "phmm_forward_chain<16,19>")
    assert e and e["hbm_bytes_per_launch"] == 1 and why is None
    assert bench.pmc_entry("config2", 1024, "phmm_forward_chain<16,25>")[0] is None
    assert bench.pmc_entry("config2", 512, "phmm_forward_chain<16,19>")[0] is None
    e, why = bench.pmc_entry("smith_waterman", 131072, "phmm_sw_align_kernel", "i32")
    assert e is None and "stale" in why


def test_committed_pmc_entries_have_the_keys_bench_reads():
    entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert entries
    for e in entries:
        for k in ("workload", "regions", "precision", "kernel_short", "src_hash", "hbm_bytes_per_launch", "valu_insts_per_launch"):
            assert k in e, (e.get("workload"), k)
