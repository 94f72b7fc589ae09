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


def test_the_compact_line_holds_the_contract_and_every_row_in_four_kilobytes():
    """what the driver stores of stdout must carry config 3 / 5, ragged, f32-first, the single region and
    the region-call rates -- numbers only, the prose lives in profiles/BENCH_NOTES.md.  Built here from a committed full
    record (round 3's), with the rows that are new this round filled in."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_final_bench.json")))
    point = {"regions_per_s": 12345, "gcups_incl_pcie": 1234.5, "us_per_call": 123.4}
    for k in ("region_call_small_30x3_1_thread", "region_call_small_30x3_8_threads_shared_handle", "region_call_small_30x3_32_threads_shared_handle",
              "region_call_ragged_1_thread", "region_call_ragged_8_threads_shared_handle", "region_call_ragged_32_threads_shared_handle"):
        full["host_calls"][k] = dict(point)
    full["smith_waterman"]["valu_issue"] = {"achieved": 0.4812, "peak": 1.2, "frac": 0.401, "valu_per_cell": 15.5, "mix_ceiling": 0.7,
                                            "frac_of_mix_ceiling": 0.6874}
    full["config"]["per_rank_cells"] = [47185920000] * 8
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and "workload" in line["config"]
    assert line["config"]["imbalance"] == 1.0 and len(line["config"]["per_rank_cells"]) == 8
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    rows = line["rows"]
    assert rows["config3_10k"]["gcups"] > 0 and rows["config5_256"]["gcups"] > 0 and rows["ragged"]["host_gcups"] > 0
    assert rows["f32_first"]["gcups"] > 0 and rows["single_region_us"] > 0
    hc = rows["host_calls"]
    assert hc["region_1t"] > 0 and hc["region_8t_shared"] > 0 and hc["small_1t"] == 12345 and hc["ragged_8t_shared"] == 12345
    assert rows["smith_waterman"]["issue_peak"] == 1.2 and rows["smith_waterman"]["mix_ceiling"] == 0.7
    # numbers only: no string in `rows` but kernel-free booleans / error texts
    def strings(x):
        if isinstance(x, dict):
            return [s for v in x.values() for s in strings(v)]
        return [x] if isinstance(x, str) else []
    assert strings(rows) == []
    assert os.path.exists(os.path.join(ROOT, line["notes"]))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_the_n_gt_1_path_of_the_bench_runs_at_world_size_two_on_one_device():
    """the driver's GPU box has one device, so the N > 1 code of bench.py (per-rank seeds, the gather of
    per-rank cells, max-over-ranks timing, rank 0 alone printing) would otherwise never execute under GPUTEST.  Two ranks over
    gloo share the device (`devices_aliased` says so): not a scaling measurement, the code path."""
    import subprocess
    import sys
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", TMPDIR="/tmp")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--regions", "256",
           "--main-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["devices_aliased"] is True
    prc = line["config"]["per_rank_cells"]
    assert len(prc) == 2 and prc[0] == prc[1] == line["config"]["cells_per_gpu_per_step"]
    # whole-job value: both ranks' cells over the slowest rank's time
    assert abs(line["value"] - sum(prc) * 3 / (line["ms_per_step"] * 3e-3) / 1e9) <= 0.01 * line["value"]
    assert line["roofline"]["bound"] == "valu_f64" and line["roofline"]["hbm"]["unit"] == "GB/s"
