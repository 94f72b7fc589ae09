#!/usr/bin/env python
"""bench.py -- PairHMM cell-updates/s (GCUPS) on MI355X, the metric BASELINE.json names.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (phmm_batch_launch: the forward kernels, nothing else) over one batch of
synthetic assembly regions already resident in HBM.

`value` (every N): per GPU `--regions` (default 1024) regions of the BASELINE.json configs[1] shape -- 128 reads x 8
haplotypes, 150 bp reads, 300 bp haplotypes -- i.e. SURVEY.md 8(d)'s batched form of config 2; rank r draws its own
regions (seed base+r): regions shard across GPUs with no collective, "weak" scaling.  Rank 0 prints ONE JSON line.

Rows next to it on the same line (BASELINE.json configs[2..4]; all ranks take part, rank 0 reports):
  config3_10k   the SAME 10 000-region set (seed 20250928, read lengths mixed {100,150,250}) at every N, sharded over the
                N ranks in contiguous ranges balanced by cells (no collective): strong scaling -- regions/s, GCUPS,
                per-rank cells, imbalance, dominant kernel and its launch time, oracle sample check
  config5_256   the same for the 256 stress regions (512 reads x 64 haplotypes, H = 400; seed 7000)
  ragged        a long-tailed mix of regions (3 ... 5 000 reads, 1 ... 128 haplotypes, H 60 ... 500, R 30 ... 250), rank 0
  f32_first     the opt-in PHMM_FLAG_F32_FIRST mode on the `value` batch (never `value`)
  single_region / engine_call / host_calls   one region per launch; the engine-level call; the reference's call
                granularity from host threads (PCIe included)

Objects:
  roofline      algorithmic HBM bytes (5*sum R + sum H + 8*Nr*Nh per region) per launch / the dominant
                kernel's mean launch duration (HIP events on the launch stream), vs 8 TB/s.  `traffic`, `l2_hit_rate`
                and `valu_issue` come from rocprofv3 PMC passes of this very command committed in
                profiles/pmc_traffic.json, and ONLY if that entry was measured on the same kernel built from the same
                sources (kernel name + hash of lorikeet_amd/csrc/*.hip,*.hpp); otherwise null and a note.
                The path is NOT HBM-bound: `valu_f64` carries the binding roofline next to it.
  cpu_baseline  the CPU oracle -- C port of the reference's scalar path -- on the host cores of this box, rank 0 at
                N=1 only, bounded sample.  `cpu_baseline_simd`: the stand-in for the reference's vector arm (gkl), our
                restatement: f32 first with f64 redo, one SIMD lane per pair (oracle/pairhmm_simd.c).
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SW_CELL_CEILING_GCUPS = 2570.0  # tools/ubench/sw_cell.hip: K = 10, 4-5 waves per SIMD (profiles/r02_ubench_sw_cell.txt)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_F64_PEAK_TFLOPS = 78.6  # vector FP64 peak (256 CU * 2.4 GHz * 128 flop/clk)
VALU_F32_PEAK_TFLOPS = 157.3
NUM_SIMD = 1024              # 256 CUs x 4
VALU_ISSUE_PEAK = 0.6        # G wave64 VALU instructions / s / SIMD at 2.4 GHz, one per 4 clk
VALU_ISSUE_PEAK_32 = 1.2     # ... for 32-bit VALU instructions: one per 2 clk (the hardware rate; what the aligner is priced against)
SW_MIX_ISSUE_CEILING = 0.70  # ... what the aligner's own 16-instruction cell can reach: nine of its instructions issue every 2.55 clk,
                             # seven (max3, alignbit, compares) every 4.3-4.8 (profiles/r02_ubench_rates.txt) -> 3.4 clk on average
VALU_ISSUE_UBENCH = 0.595    # what the f64 kernel's own cell body sustains alone (tools/ubench/issue.hip, 2 waves/SIMD)
FLOP_PER_CELL = 12           # M 4 mul + 2 add, I 2+1, D 2+1
PAYLOAD = ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=None, help="regions per GPU per step of the main workload")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5", "ragged"])
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: the timed loop IS BASELINE.json configs[2]/[3] -- the same 10 000 regions at every N, sharded "
                         "over the ranks in contiguous cell-balanced ranges -- and `value` is its aggregate rate (at N = 1 it is "
                         "the config3_10k row of the default line)")
    ap.add_argument("--f32-first", action="store_true", help="PMC passes only: the main loop on a PHMM_FLAG_F32_FIRST engine")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flush-caches", action="store_true",
                    help="PMC calibration only: overwrite a 1 GiB buffer before every launch so that no input byte "
                         "survives in L2 / Infinity Cache from the previous launch (the timing then includes the fill)")
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed loop (no extra rows, no cpu_baseline): for PMC passes")
    return ap.parse_args()


def make_workload(name, n_regions, seed):
    from lorikeet_amd import synthetic
    if name == "config2":
        return synthetic.config2(n_regions or 1024, seed=seed), "128 reads x 8 haps, R=150, H=300"
    if name == "config3":
        return synthetic.config3(n_regions or 10000), "128 reads x 8 haps, H=300, R in {100,150,250}"
    if name == "config5":
        return synthetic.config5(n_regions or 256), "512 reads x 64 haps, R=150, H=400"
    return synthetic.ragged(n_regions or 1536), "long-tailed mix: 3..5000 reads x 1..128 haps, H 60..500, R 30..250"


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container that
    sees 256 logical CPUs may be limited to 16 cores' worth of time; more threads than that only oversubscribe)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except Exception:
            pass
    return n


def cpu_baselines(batch, budget_s=12.0):
    """(scalar port, SIMD stand-in) on all usable host cores, each on a bounded sample of rank 0's batch."""
    from oracle import oracle
    cores = usable_cores()
    try:
        oracle.build(native=True)
        native = True
    except Exception:
        native = False
    cpu = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass

    def timed(fn, n0):
        sub = batch.region_slice(0, n0)
        t = time.perf_counter()
        fn(sub)
        dt0 = time.perf_counter() - t
        rounds = max(1, min(int(budget_s / max(dt0, 1e-3)), batch.n_regions // n0))
        sub = batch.region_slice(0, n0 * rounds)
        t = time.perf_counter()
        res = fn(sub)
        return sub, time.perf_counter() - t, res

    n0 = min(batch.n_regions, cores)
    sub, dt, want = timed(lambda s: oracle.compute_batch(s.as_dict(), n_threads=cores, native=native), n0)
    scalar = {"value": round(sub.cells() / dt / 1e9, 4), "unit": "GCUPS", "cores": cores, "kind": "port",
              "sample": "first %d regions of rank 0's batch (%.3g cells, %.1f s); oracle/pairhmm_oracle.c, the reference's "
                        "SCALAR arm (f64), one region per pthread task%s; cores = min(affinity, cgroup cpu quota) of %d "
                        "logical CPUs; %s" % (sub.n_regions, sub.cells(), dt, ", -march=native" if native else "",
                                              os.cpu_count() or 1, cpu)}
    simd = None
    try:
        sub2, dt2, (got, redone) = timed(lambda s: oracle.compute_batch_simd(s.as_dict(), n_threads=cores, native=native),
                                         min(batch.n_regions, 4 * cores))
        n = min(sub.n_out, sub2.n_out)
        simd = {"value": round(sub2.cells() / dt2 / 1e9, 4), "unit": "GCUPS", "cores": cores, "kind": "port",
                "stand_in_for": "gkl::pairhmm::forward (the reference's default AVX arm, pair_hmm.rs:348-366) -- OUR "
                                "restatement, not gkl: the crate is not vendored and cannot be built here",
                "sample": "first %d regions (%.3g cells, %.1f s); oracle/pairhmm_simd.c: f32 under 2^120 with f64 redo "
                          "below 1e-28, one SIMD lane per (read, haplotype) pair, %d lanes (gcc vector extensions%s), one "
                          "region per pthread task" % (sub2.n_regions, sub2.cells(), dt2, oracle.lib().oracle_simd_lanes(),
                                                       ", -march=native" if native else ""),
                "pairs_redone_in_f64": redone, "max_abs_diff_vs_scalar_port": float(abs(got[:n] - want[:n]).max()),
                "tolerance": 1e-5}
    except Exception as exc:  # an optional row must never cost the bench line
        simd = {"error": repr(exc)}
    return scalar, simd


sys.path.insert(0, os.path.join(ROOT, "tools"))
from source_hash import KERNEL_SOURCES, source_hash  # noqa: E402  (shared with the Makefile: phmm_build_info() carries the same hashes)


def pmc_entry(workload, regions, kernel, precision="f64"):
    """The rocprofv3 PMC measurement of this command (profiles/pmc_traffic.json; tools/profile.sh +
    tools/pmc_update.py), or (None, why) when there is none for this workload, kernel and source hash."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        entries = json.load(open(path))
    except Exception:
        return None, "profiles/pmc_traffic.json missing"
    src = source_hash("sw" if workload == "smith_waterman" else "pairhmm")
    why = "no PMC entry for workload %s x %s (%s)" % (workload, regions, precision)
    for e in entries:
        if e.get("workload") != workload or e.get("regions") != regions or e.get("precision", "f64") != precision:
            continue
        if e.get("kernel_short") != kernel:
            why = "PMC entry is for kernel %s, this run's dominant kernel is %s" % (e.get("kernel_short"), kernel)
            continue
        if e.get("src_hash") != src:
            why = "PMC entry was measured on kernel sources %s, this tree is %s: stale, not reported" % (e.get("src_hash"), src)
            continue
        return e, None
    return None, why


class Dist:
    """barrier + max-over-ranks + tiny gathers only; no data-path collective."""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != a.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, self.world))
        # one process per GPU; BENCH_DIST_BACKEND=gloo (ranks may then share a device) exists only to exercise the
        # N>1 code path on a single-GPU box
        self.backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        self.dev_index = self.local_rank if self.backend == "nccl" else self.local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        if self.world > 1:
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier(device_ids=[self.dev_index]) if self.backend == "nccl" else self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank (one float64 each)."""
        if self.world == 1:
            return [float(x)]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        t[self.rank] = x
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def close(self):
        if self.world > 1:
            self.barrier()
            self.dist.destroy_process_group()


class Resident:
    """A batch resident in HBM, bound to a launch plan."""

    def __init__(self, eng, batch, dev):
        import torch
        self.batch = batch
        self.plan = eng.plan(batch)
        self.tens = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in PAYLOAD}
        self.out = torch.empty(batch.n_out, dtype=torch.float64, device=dev)
        self.plan.bind_torch(self.tens, self.out)

    def close(self):
        self.plan.close()


def timed_launches(D, res, stream, steps, warmup, flush=None):
    """`steps` launches bracketed by barrier + synchronize on both sides; returns (max-over-ranks seconds, per-launch ms
    from HIP events on the launch stream)."""
    torch = D.torch
    sh = stream.cuda_stream
    with torch.cuda.stream(stream):
        for _ in range(warmup):
            res.plan.launch(sh)
        D.barrier()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        ev[0].record(stream)
        for i in range(steps):
            if flush is not None:
                flush.fill_(float(i))
            res.plan.launch(sh)
            ev[i + 1].record(stream)
        D.barrier()
        elapsed = time.perf_counter() - t0
    res.plan.status()  # raises if any likelihood came out > 0
    return D.max(elapsed), [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def oracle_sample_diff(batch, got, max_regions, max_cells):
    """max |hip - oracle| over the first regions of `batch` that fit the CPU budget (bench rows are checked, not only
    `<= 0`)."""
    import numpy as np
    from lorikeet_amd import sharding
    from oracle import oracle
    cells = sharding.region_cells(batch)
    picked, used = [], 0
    for g in range(batch.n_regions):
        if len(picked) >= max_regions:
            break
        if used + int(cells[g]) <= max_cells:
            picked.append(g)
            used += int(cells[g])
    worst = 0.0
    for g in picked:
        sub = batch.region_slice(g, g + 1)
        want = oracle.compute_batch(sub.as_dict(), n_threads=usable_cores())
        have = got[int(batch.out_off[g]):int(batch.out_off[g + 1])]
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(have), fin), "bench: -inf / NaN pattern differs from the oracle in region %d" % g
        if fin.any():
            worst = max(worst, float(np.max(np.abs(have[fin] - want[fin]))))
    return {"regions_checked": len(picked), "cells_checked": used, "max_abs_diff": worst, "tolerance": 1e-9}


def strong_row(D, eng, stream, name, steps, sample):
    """One fixed set at every N, sharded over the ranks in contiguous cell-balanced ranges (BASELINE.json configs[3],[4])."""
    import numpy as np
    from lorikeet_amd import sharding, synthetic
    n_regions, seed = synthetic.CONFIGS[name][4], synthetic.CONFIGS[name][5]
    cells = synthetic.config_cells(name)  # cheap: read lengths only
    bounds = sharding.split_contiguous(cells, D.world)
    lo, hi = bounds[D.rank], bounds[D.rank + 1]
    batch = synthetic.config(name, only=(lo, hi))  # every rank makes only its own regions (chunk-seeded generator)
    res = Resident(eng, batch, D.dev)
    elapsed, kern_ms = timed_launches(D, res, stream, steps, 1)
    mine = int(cells[lo:hi].sum())
    assert mine == res.plan.cells
    per_rank = D.gather(mine)
    row = None
    if D.rank == 0:
        got = res.out.cpu().numpy()
        assert (got <= 0).all(), "non-finite or positive likelihoods"
        total = float(cells.sum())
        kms = sum(kern_ms) / len(kern_ms)
        row = {"workload": "%s: the same %d regions (seed %d) at every N" % (name, n_regions, seed), "scaling": "strong",
               "regions": n_regions, "cells": int(total), "steps": steps,
               "gcups": round(total * steps / elapsed / 1e9, 1), "regions_per_s": round(n_regions * steps / elapsed, 1),
               "ms_per_step": round(elapsed / steps * 1e3, 4),
               "sharding": "contiguous ranges of regions balanced by cells (sharding.split_contiguous == "
                           "phmm_split_regions), one process per GPU, no collective",
               "per_rank_cells": [int(c) for c in per_rank],
               "imbalance": round(max(per_rank) / (sum(per_rank) / len(per_rank)), 5),
               "rank0": {"regions": hi - lo, "kernel": res.plan.dominant_kernel, "launches": res.plan.num_launches,
                         "kernel_ms": round(kms, 4), "gcups": round(mine / (sum(kern_ms) / len(kern_ms)) / 1e6, 1),
                         "algorithmic_bytes_per_launch": res.plan.algorithmic_bytes,
                         "hbm_achieved_gbs": round(res.plan.algorithmic_bytes / kms / 1e6, 3)},
               "oracle_sample": oracle_sample_diff(batch, got, *sample)}
        row["valu_f64"] = {"achieved": round(FLOP_PER_CELL * mine / kms / 1e9, 3), "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(FLOP_PER_CELL * mine / kms / 1e9 / VALU_F64_PEAK_TFLOPS, 4), "flop_per_cell": FLOP_PER_CELL,
                           "note": "rank 0's shard: 12 flop per cell of the reference recurrence x its cells / its mean launch time"}
        e, why = pmc_entry(name, n_regions, res.plan.dominant_kernel) if D.world == 1 else (None, "PMC entries are for N=1")
        row["rocprof"] = ({"hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "l2_hit_rate": e.get("l2_hit_rate"),
                           "hbm_gbs_from_counters": round(e["hbm_bytes_per_launch"] / kms / 1e6, 2),
                           "traffic_over_algorithmic": round(e["hbm_bytes_per_launch"] / res.plan.algorithmic_bytes, 3),
                           "valu_per_cell": round(e["valu_insts_per_launch"] * 64 / mine, 3) if e.get("valu_insts_per_launch") else None,
                           "source": e.get("source")} if e else {"hbm_bytes_per_launch": None, "note": why})
    if D.world == 1 and D.rank == 0 and row is not None:  # the reference's production arithmetic (gkl: f32 first) on the same set
        try:
            from lorikeet_amd import HipPairHMMEngine
            torch = D.torch
            e32 = HipPairHMMEngine(D.dev_index, f32_first=True)
            p32 = e32.plan(batch)
            out32 = torch.empty(batch.n_out, dtype=torch.float64, device=D.dev)
            p32.bind_torch(res.tens, out32)
            sh = stream.cuda_stream
            with torch.cuda.stream(stream):
                p32.launch(sh)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(steps):
                    p32.launch(sh)
                e1.record(stream)
                stream.synchronize()
            p32.status()
            ms32 = e0.elapsed_time(e1) / steps
            row["f32_first"] = {"gcups": round(p32.cells / ms32 / 1e6, 1), "ms_per_step": round(ms32, 4), "kernel": p32.dominant_kernel,
                                "max_abs_diff_vs_f64": float((out32 - res.out).abs().max().item()), "tolerance": 1e-5,
                                "valu_f32": {"achieved": round(FLOP_PER_CELL * p32.cells / ms32 / 1e9, 3), "peak": VALU_F32_PEAK_TFLOPS,
                                             "unit": "TFLOP/s", "frac": round(FLOP_PER_CELL * p32.cells / ms32 / 1e9 / VALU_F32_PEAK_TFLOPS, 4)},
                                "note": "opt-in PHMM_FLAG_F32_FIRST on the same resident set; never the row's gcups"}
            p32.close()
            e32.close()
        except Exception as exc:  # an optional sub-row must never cost the row
            row["f32_first"] = {"error": repr(exc)}
    res.close()
    return row


def write_full_record(line):
    """Everything the run measured, with its explanatory notes, as a file (stdout carries the compact line below)."""
    out_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.environ.get("TMPDIR", "/tmp")
    path = os.path.join(out_dir, "bench_full.json")
    try:
        with open(path, "w") as f:
            json.dump(line, f, indent=1)
        return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError:
        return None


def compact_line(line, full_path):
    """The bench line the driver records: the contract's fields, `roofline`, `cpu_baseline`, and ONE numbers-only object
    `rows` for every other measurement of the run -- under 4 KB, so that the stored tail of stdout holds all of it.  What each
    number is, how it was measured and what it is priced against: profiles/BENCH_NOTES.md (field by field)."""
    def g(d, *path, nd=None):
        for k in path:
            if not isinstance(d, dict) or d.get(k) is None:
                return None
            d = d[k]
        return round(d, nd) if nd is not None and isinstance(d, float) else d

    def pick(d, **fields):  # {short name: path or key}; rows that failed keep their error text
        if not isinstance(d, dict):
            return None
        if "error" in d:
            return {"error": str(d["error"])[:80]}
        return {k: g(d, *(v if isinstance(v, tuple) else (v,))) for k, v in fields.items()}

    c = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "devices_aliased", "dtype", "data")}
    cfg = line["config"]
    prc = cfg.get("per_rank_cells") or []
    c["config"] = {"workload": cfg["workload"].split(":")[0] if line["scaling"] == "strong" else cfg["workload"].split(" (")[0],
                   "regions_per_gpu": cfg["regions_per_gpu"], "cells_per_gpu_per_step": cfg["cells_per_gpu_per_step"], "seed": cfg["seed"],
                   "per_rank_cells": prc if len(prc) <= 8 else None,
                   "imbalance": round(max(prc) * len(prc) / max(sum(prc), 1), 4) if prc else None, "sharding": "regions; no collective",
                   "env": cfg.get("env")}
    c["regions_per_s"] = line["regions_per_s"]
    c["roofline"] = pick(line["roofline"], bound="bound", achieved="achieved", peak="peak", unit="unit", frac="frac", traffic="traffic",
                         l2_hit_rate="l2_hit_rate", kernel="kernel", kernel_ms="kernel_ms",
                         algorithmic_bytes_per_launch="algorithmic_bytes_per_launch", src_hash="src_hash")
    if isinstance(line["roofline"].get("hbm"), dict):   # (the HBM figure the metric's wording asks for, beside the bound that binds)
        c["roofline"]["hbm"] = pick(line["roofline"]["hbm"], achieved="achieved", peak="peak", unit="unit", frac="frac")
    c["roofline"]["definition"] = "r5+: the bound that binds (%s); the HBM figure of r1-r4 lines is roofline.hbm" % line["roofline"]["bound"]
    f32 = line.get("f32_first")
    if isinstance(f32, dict) and "error" not in f32:  # (the arithmetic the reference ships -- gkl, f32 first -- priced like the headline)
        c["roofline_f32"] = {"bound": "valu_f32", "achieved": g(f32, "valu_f32", "achieved"), "peak": g(f32, "valu_f32", "peak"), "unit": "TFLOP/s",
                             "frac": g(f32, "valu_f32", "frac"), "kernel": g(f32, "kernel"), "kernel_ms": g(f32, "ms_per_step"),
                             "issue_frac": g(f32, "valu_issue", "frac"), "valu_per_cell": g(f32, "valu_issue", "valu_per_cell"),
                             "traffic": g(f32, "valu_issue", "hbm_bytes_per_launch"), "l2_hit_rate": g(f32, "valu_issue", "l2_hit_rate")}
    c["valu_f64"] = pick(line["valu_f64"], achieved="achieved", peak="peak", unit="unit", frac="frac")
    c["valu_issue"] = pick(line["valu_issue"], achieved="achieved", peak="peak", frac="frac", valu_per_cell="valu_per_cell")
    for k in ("cpu_baseline", "cpu_baseline_simd"):
        if line.get(k):
            c[k] = pick(line[k], value="value", unit="unit", cores="cores", kind="kind", sample="sample")
            if isinstance(c[k].get("sample"), str):
                c[k]["sample"] = c[k]["sample"].split(";")[0][:90]  # (what was timed; the rest is in the full record)
    strong = lambda r: pick(r, gcups="gcups", ms="ms_per_step", regions_per_s="regions_per_s", valu_f64_frac=("valu_f64", "frac"),  # noqa: E731
                            imbalance="imbalance", f32_first_gcups=("f32_first", "gcups"), max_abs_diff=("oracle_sample", "max_abs_diff"))
    hc = line.get("host_calls") or {}
    rps = lambda k: g(hc, k, "regions_per_s")  # noqa: E731
    sw = line.get("smith_waterman")
    rl = line.get("realign_to_best")
    rows = {
        "config3_10k": strong(line.get("config3_10k")), "config5_256": strong(line.get("config5_256")),
        "ragged": pick(line.get("ragged"), gcups="gcups", ms="ms_per_step", regions_per_s="regions_per_s", launches="launches_per_step",
                       executed_per_cell="executed_per_cell",
                       host_gcups=("host_buffers_incl_pcie", "gcups"), host_ms=("host_buffers_incl_pcie", "ms_per_call"),
                       valu_f64_frac=("valu_f64", "frac"), max_abs_diff=("oracle_sample", "max_abs_diff")),
        "f32_first": pick(line.get("f32_first"), gcups="value", ms="ms_per_step", max_abs_diff="max_abs_diff_vs_f64"),
        "single_region_us": g(line, "single_region", "us_per_region"),
        "engine_call": pick(line.get("engine_call"), gcups_incl_pcie="gcups_incl_pcie", ms="ms_per_call", regions="regions"),
        # regions/s through host buffers, one region per call unless the name says otherwise (tools/threads_bench)
        "host_calls": None if not hc else ({"error": str(hc["error"])[:80]} if "error" in hc else {
            "pairhmm_8t_own": rps("one_region_per_call_8_threads_own_handles"), "pairhmm_32t_shared": rps("one_region_per_call_32_threads_shared_handle_submit_wait"),
            "region_1t": rps("region_call_one_region_per_call_1_thread"), "region_1t_us": g(hc, "region_call_one_region_per_call_1_thread", "us_per_call"),
            "region_4t_own": rps("region_call_one_region_per_call_4_threads_own_handles"),
            "region_8t_own": rps("region_call_one_region_per_call_8_threads_own_handles"),
            "region_8t_shared": rps("region_call_one_region_per_call_8_threads_shared_handle"),
            "region_10t_shared": rps("region_call_one_region_per_call_10_threads_shared_handle"),
            "region_16t_shared": rps("region_call_one_region_per_call_16_threads_shared_handle"),
            "region_32t_shared": rps("region_call_one_region_per_call_32_threads_shared_handle"),
            "region_8t_shared_2tk": rps("region_call_one_region_per_call_8_threads_shared_handle_2_tickets"),
            "region_10t_shared_2tk": rps("region_call_one_region_per_call_10_threads_shared_handle_2_tickets"),
            "region_16t_shared_2tk": rps("region_call_one_region_per_call_16_threads_shared_handle_2_tickets"),
            "region_10t_own": rps("region_call_one_region_per_call_10_threads_own_handles"),
            "region_16t_own": rps("region_call_one_region_per_call_16_threads_own_handles"),
            "region_32t_own": rps("region_call_one_region_per_call_32_threads_own_handles"),
            "pairhmm_16t_own": rps("one_region_per_call_16_threads_own_handles"),
            "pairhmm_32t_own": rps("one_region_per_call_32_threads_own_handles"),
            "region_4t_x8": rps("region_call_eight_regions_per_call_4_threads_own_handles"), "region_1t_x64": rps("region_call_64_regions_per_call_1_thread"),
            "small_1t": rps("region_call_small_30x3_1_thread"), "small_1t_us": g(hc, "region_call_small_30x3_1_thread", "us_per_call"),
            "small_8t_shared": rps("region_call_small_30x3_8_threads_shared_handle"), "small_32t_shared": rps("region_call_small_30x3_32_threads_shared_handle"),
            "ragged_1t": rps("region_call_ragged_1_thread"), "ragged_8t_shared": rps("region_call_ragged_8_threads_shared_handle"),
            "two_calls_1t": rps("likelihoods_then_realignment_one_region_per_call_1_thread")}),
        "smith_waterman": pick(sw, gcups_i32="gcups_i32", ms="ms_per_call", kernel_gcups_i32=("kernel", "gcups_i32"), kernel_ms=("kernel", "ms"),
                               one_piece_gcups_i32=("kernel_one_piece", "gcups_i32"), full_instance_gcups_i32=("full_instance_only", "kernel_gcups_i32"),
                               issue=("valu_issue", "achieved"), issue_peak=("valu_issue", "peak"), issue_frac=("valu_issue", "frac"),
                               mix_ceiling=("valu_issue", "mix_ceiling"), frac_of_mix_ceiling=("valu_issue", "frac_of_mix_ceiling"),
                               valu_per_cell=("valu_issue", "valu_per_cell"), algorithmic_bytes=("roofline", "algorithmic_bytes_per_launch"),
                               traffic=("roofline", "traffic"), flag_bytes=("roofline", "backtrack_flag_bytes_per_launch"),
                               equal_to_oracle="equal_to_oracle_on_sample", cpu_gcups_i32=("cpu_oracle", "gcups_i32"),
                               indel_reads_gcups_i32=("indel_rich", "reads_150_with_2_to_5_percent_indels", "gcups_i32"),
                               indel_haps_gcups_i32=("indel_rich", "haplotype_to_reference_400x400", "gcups_i32")),
        "realign": pick(rl, ms="ms_per_call", reads_per_s="reads_per_s", project_ms=("project_to_reference", "ms_per_call"),
                        one_call_ms=("realign_reads_one_call", "ms_per_call"), equal_to_oracle="equal_to_oracle_on_sample",
                        project_equal_to_oracle=("project_to_reference", "equal_to_oracle_on_sample")),
        "oracle_sample_max_abs_diff": g(line, "oracle_sample", "max_abs_diff"),
    }
    c["rows"] = {k: v for k, v in rows.items() if v is not None}
    c["notes"] = "profiles/BENCH_NOTES.md"
    c["full_record"] = full_path
    return c


def main():
    a = parse()
    # This process launches whole batches on streams of its own choosing; the host_calls rows are measured in child processes
    # (tools/threads_bench).  Idle hardware queues of ANOTHER process are not free for those (a second process holding two
    # engines with their queue pool: 115.7 -> 120.7 us per region call), so this one keeps none: its engines stay on ordinary
    # streams, the children run with the library's defaults.
    own_queue_given = "PHMM_REGION_OWN_QUEUE" in os.environ
    os.environ.setdefault("PHMM_REGION_OWN_QUEUE", "0")
    D = Dist(a)
    torch, dev, rank, world = D.torch, D.dev, D.rank, D.world
    extras = not a.main_only

    from lorikeet_amd import HipPairHMMEngine
    strong = None
    if a.scaling == "strong":  # the fixed set of BASELINE.json configs[2]/[3]; every rank makes only its own regions
        from lorikeet_amd import sharding, synthetic
        a.workload = "config3"
        all_cells = synthetic.config_cells("config3")
        bounds = sharding.split_contiguous(all_cells, world)
        batch = synthetic.config("config3", only=(bounds[rank], bounds[rank + 1]))
        shape = "128 reads x 8 haps, H=300, R in {100,150,250}"
        strong = {"regions": synthetic.CONFIGS["config3"][4], "seed": synthetic.CONFIGS["config3"][5], "cells": float(all_cells.sum())}
    else:
        batch, shape = make_workload(a.workload, a.regions, a.seed + rank)
    regions = batch.n_regions
    eng = HipPairHMMEngine(D.dev_index, f32_first=a.f32_first)
    res = Resident(eng, batch, dev)
    plan, out, tens = res.plan, res.out, res.tens
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream

    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev) if a.flush_caches else None
    elapsed, kern_ms = timed_launches(D, res, stream, a.steps, a.warmup, flush)
    cells_total = plan.cells * world  # identical shapes on every rank
    regions_total = regions * world
    per_rank_cells = [int(c) for c in D.gather(plan.cells)]  # (every rank takes part)
    if strong:
        cells_total, regions_total = strong["cells"], strong["regions"]

    def optional(fn):
        try:
            return fn()
        except Exception as exc:  # an optional row must never cost the bench line
            return {"error": repr(exc)}

    # ---- rows every rank takes part in: the fixed sets of BASELINE.json configs[2..4], strong scaling -------------
    config3_row = config5_row = None
    if extras and a.workload == "config2" and not strong and (not os.environ.get("BENCH_ROWS") or "config35" in os.environ["BENCH_ROWS"].split(",")):
        try:
            config3_row = strong_row(D, eng, stream, "config3", 5, (2, int(2e8)))
        except Exception as exc:
            config3_row = {"error": repr(exc)}
        try:
            config5_row = strong_row(D, eng, stream, "config5", 5, (1, int(2.1e9)))
        except Exception as exc:
            config5_row = {"error": repr(exc)}

    def single_region():  # configs[1] literally: ONE region per launch (latency mode: the planner spreads it over all SIMDs)
        one, _ = make_workload("config2", 1, a.seed + 7919)
        r1 = Resident(eng, one, dev)
        with torch.cuda.stream(stream):
            for _ in range(10):
                r1.plan.launch(sh)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(100):
                r1.plan.launch(sh)
            e1.record(stream)
            stream.synchronize()
        us = e0.elapsed_time(e1) * 10.0
        row = {"regions": 1, "us_per_region": round(us, 2), "gcups": round(r1.plan.cells / us / 1e3, 1),
               "kernel": r1.plan.dominant_kernel, "note": "one region per launch, back-to-back launches on one stream"}
        r1.close()
        return row

    def f32_first():  # the opt-in PHMM_FLAG_F32_FIRST mode on the same resident batch (never `value`)
        e32 = HipPairHMMEngine(D.dev_index, f32_first=True)
        p32 = e32.plan(batch)
        out32 = torch.empty(batch.n_out, dtype=torch.float64, device=dev)
        p32.bind_torch(tens, out32)
        with torch.cuda.stream(stream):
            for _ in range(2):
                p32.launch(sh)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(a.steps):
                p32.launch(sh)
            e1.record(stream)
            stream.synchronize()
        p32.status()
        ms32 = e0.elapsed_time(e1) / a.steps
        row = {"value": round(p32.cells / ms32 / 1e6, 2), "unit": "GCUPS", "ms_per_step": round(ms32, 4),
               "kernel": p32.dominant_kernel, "dtype": "f32 first, f64 redo of what f32 cannot be trusted with",
               "max_abs_diff_vs_f64": float((out32 - out).abs().max().item()), "tolerance": 1e-5,
               "valu_f32": {"achieved": round(FLOP_PER_CELL * p32.cells / ms32 / 1e9, 3), "peak": VALU_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(FLOP_PER_CELL * p32.cells / ms32 / 1e9 / VALU_F32_PEAK_TFLOPS, 4)},
               "note": "opt-in flag of phmm_create, what the reference's vector arm (gkl) does; default and `value` are f64"}
        e, why = pmc_entry(a.workload, regions, p32.dominant_kernel, "f32_first")
        if e and e.get("valu_insts_per_launch"):
            rate = e["valu_insts_per_launch"] / (ms32 / 1e3) / NUM_SIMD / 1e9
            row["valu_issue"] = {"achieved": round(rate, 4), "peak": 2 * VALU_ISSUE_PEAK, "unit": "G wave64-instr/s per SIMD",
                                 "frac": round(rate / (2 * VALU_ISSUE_PEAK), 4), "same_mix_ubench": 1.11,
                                 "note": "32-bit VALU instructions issue every 2 clk (peak 1.2 G/s per SIMD at 2.4 GHz); the f32 cell "
                                         "body alone sustains 1.11 (tools/ubench/issue.hip, salu_mask.hip)",
                                 "valu_per_cell": round(e["valu_insts_per_launch"] * 64 / p32.cells, 3),
                                 "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "l2_hit_rate": e.get("l2_hit_rate"),
                                 "source": e.get("source")}
        else:
            row["valu_issue"] = {"achieved": None, "note": why}
        p32.close()
        e32.close()
        return row

    def engine_call():  # SURVEY 8(f1/f2): the engine-level call (pre-step + PairHMM + normalise/disqualify), host buffers
        import ctypes as C
        import math
        import numpy as np
        from lorikeet_amd import _lib
        nreg = min(256, batch.n_regions)
        sub = batch.region_slice(0, nreg)
        cfg = _lib.EngineConfig()
        cfg.constant_gcp, cfg.pcr_error_model, cfg.base_quality_score_threshold = 10, 3, 18
        cfg.dynamic_read_disqualification, cfg.symmetrically_normalize_alleles_to_reference = 1, 1
        cfg.log10_global_read_mismapping_rate = -4.5 * math.log10(math.e)
        cfg.read_disqualification_scale, cfg.expected_error_rate_per_base = 1.0, 0.02
        mapq = np.full(sub.n_reads, 60, np.uint8)
        ref = np.zeros(nreg, np.int32)
        eout = np.empty(sub.n_out, np.float64)
        keep = np.zeros(sub.n_reads, np.uint8)
        pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
        args = (eng._h, C.byref(cfg), nreg, pp(sub.region_read_off, _lib.u32p), pp(sub.region_hap_off, _lib.u32p),
                pp(sub.read_off, _lib.u32p), pp(sub.read_bases, _lib.u8p), pp(sub.base_q, _lib.u8p), None, None,
                pp(mapq, _lib.u8p), pp(sub.hap_off, _lib.u32p), pp(sub.hap_bases, _lib.u8p),
                pp(ref, C.POINTER(C.c_int32)), pp(sub.out_off, _lib.u64p), pp(eout, _lib.f64p), pp(keep, _lib.u8p))
        assert eng.lib.phmm_engine_compute(*args) == 0, eng.last_error()
        te = time.perf_counter()
        for _ in range(5):
            assert eng.lib.phmm_engine_compute(*args) == 0
        te = (time.perf_counter() - te) / 5
        return {"call": "phmm_engine_compute (PCR model conservative, dynamic disqualification), host buffers, "
                        "PCIe included", "regions": nreg, "ms_per_call": round(te * 1e3, 3),
                "gcups_incl_pcie": round(sub.cells() / te / 1e9, 1), "reads_kept_fraction": round(float(keep.mean()), 4)}

    def host_calls():
        # The reference's call granularity (one region per compute_likelihoods call from every rayon worker), host buffers,
        # PCIe included: tools/threads_bench (C++ threads on the C ABI, built with the library) in three configurations.
        import re
        import subprocess
        exe = os.path.join(ROOT, "tools", "threads_bench")

        def point(mode, threads, per_call, shape=("128", "8", "150", "300"), seconds="1.0", depth=1):
            env = dict(os.environ, TB_MODE=mode, TB_THREADS=str(threads), TB_DEPTH=str(depth))
            if not own_queue_given:
                env.pop("PHMM_REGION_OWN_QUEUE", None)
            if shape == "ragged":
                env["TB_SHAPE"] = "ragged"
                shape = ("128", "8", "150", "300")
            # (many threads on one handle: who leads which flush is the scheduler's business, a point moves by +-10 % from run
            # to run -- the median of three shorter runs is quoted)
            # (windows of a full second: a point of 0.36 s reads 26-44 k regions/s where 1 s reads 50 k -- 30 x 3 regions from 8
            # threads, same box: the device's clocks and the combiner's flush sizes take a few tenths of a second to settle)
            runs = []
            for _ in range(3 if threads >= 8 else 1):
                secs = "%.2f" % max(float(seconds), 1.0 if threads >= 8 else 0.0)
                r = subprocess.run([exe, secs, *shape, str(per_call)], env=env, capture_output=True, text=True, timeout=120)
                m = re.search(r"threads:\s+(\d+) regions/s\s+([\d.]+) GCUPS\s+([\d.]+) us per call", r.stdout)
                runs.append({"regions_per_s": int(m.group(1)), "gcups_incl_pcie": float(m.group(2)), "us_per_call": float(m.group(3))})
            return sorted(runs, key=lambda x: x["regions_per_s"])[len(runs) // 2]
        small = ("30", "3", "150", "300")  # (what most real regions look like: a few dozen reads, two or three haplotypes)
        return {"note": "config-2 regions through host buffers (PCIe, planning and staging included), C++ caller threads",
                "one_region_per_call_4_threads_own_handles": point("own", 4, 1, seconds="0.6"),
                "one_region_per_call_8_threads_own_handles": point("own", 8, 1),
                "one_region_per_call_32_threads_shared_handle_submit_wait": point("shared", 32, 1),
                "eight_regions_per_call_4_threads_own_handles": point("own", 4, 8),
                # ... and with the realignment behind it (phmm_compute, then phmm_realign_reads: best alleles, alignments,
                # projection onto the reference), as the reference's region loop goes on
                "likelihoods_then_realignment_one_region_per_call_1_thread": point("pipeline", 1, 1),
                "likelihoods_then_realignment_one_region_per_call_8_threads": point("pipeline", 8, 1),
                "likelihoods_then_realignment_eight_regions_per_call_4_threads": point("pipeline", 4, 8),
                # ... and as ONE call (round 3): phmm_region_compute / phmm_region_submit -- pre-step, PairHMM, exact pass,
                # post-step, best alleles, Smith-Waterman, projection in one enqueue, the likelihood matrix never leaving
                # the device (the engine-level pre- and post-step are work the two-call rows above do not do)
                "region_call_one_region_per_call_1_thread": point("fused", 1, 1),
                "region_call_one_region_per_call_4_threads_own_handles": point("fused", 4, 1, seconds="0.6"),
                "region_call_one_region_per_call_8_threads_own_handles": point("fused", 8, 1),
                "region_call_one_region_per_call_8_threads_shared_handle": point("gshared", 8, 1),
                # (Lorikeet's default is --threads 10, src/cli.rs:1379-1383)
                "region_call_one_region_per_call_10_threads_shared_handle": point("gshared", 10, 1),
                "region_call_one_region_per_call_16_threads_shared_handle": point("gshared", 16, 1),
                "region_call_one_region_per_call_32_threads_shared_handle": point("gshared", 32, 1),
                # ... with TWO tickets in flight per worker (region k+1 submitted before region k is waited for; what the patch's
                # rayon pool of 2 x --threads workers amounts to, integration/lorikeet-hip.patch src/bin/lorikeet.rs)
                "region_call_one_region_per_call_8_threads_shared_handle_2_tickets": point("gshared", 8, 1, depth=2),
                "region_call_one_region_per_call_10_threads_shared_handle_2_tickets": point("gshared", 10, 1, depth=2),
                "region_call_one_region_per_call_16_threads_shared_handle_2_tickets": point("gshared", 16, 1, depth=2),
                # ... and a private handle per worker past five (INTEGRATION.md section 4's thread_local!): their one-shot calls go through
                # the device's resident region server (round 6: nothing launched, results bit-reproducible; round 5 routed them through
                # the shared combiner, round 4: 22 k at 16 threads, 12.5 k at 32)
                "region_call_one_region_per_call_10_threads_own_handles": point("fused", 10, 1),
                "region_call_one_region_per_call_16_threads_own_handles": point("fused", 16, 1),
                "region_call_one_region_per_call_32_threads_own_handles": point("fused", 32, 1),
                "one_region_per_call_16_threads_own_handles": point("own", 16, 1),
                "one_region_per_call_32_threads_own_handles": point("own", 32, 1),
                "region_call_eight_regions_per_call_4_threads_own_handles": point("fused", 4, 8),
                "region_call_64_regions_per_call_1_thread": point("fused", 1, 64),
                # ... on the regions a real `lorikeet call` mostly issues: 30 reads x 3 haplotypes, and the ragged mix
                "region_call_small_30x3_1_thread": point("fused", 1, 1, small, "0.6"),
                "region_call_small_30x3_8_threads_shared_handle": point("gshared", 8, 1, small, "0.6"),
                "region_call_small_30x3_32_threads_shared_handle": point("gshared", 32, 1, small, "0.6"),
                "region_call_ragged_1_thread": point("fused", 1, 1, "ragged", "0.8"),
                "region_call_ragged_8_threads_shared_handle": point("gshared", 8, 1, "ragged", "0.8"),
                "region_call_ragged_32_threads_shared_handle": point("gshared", 32, 1, "ragged", "0.8")}

    def ragged():
        """Real regions span 3 x 2 ... 5 000 x 128: the planner on a long-tailed mix, resident and through host buffers."""
        from lorikeet_amd import sharding
        rb, rshape = make_workload("ragged", None, 0)
        r = Resident(eng, rb, dev)
        # (two rounds of five launches, the better one quoted: one round in the full run of round 6 came out at 20.5 ms where twenty
        # others on five boxes gave 15.6-15.8 -- a secondary row should not hang on one hiccup)
        el, kms = min((timed_launches(Dist1(D), r, stream, 5, 1) for _ in range(2)), key=lambda t: t[0])
        got = r.out.cpu().numpy()
        # (the first call grows the arenas; of five more the best is quoted and the median kept beside it: the chunked path's
        # time moves by +-0.5 ms from call to call with where the chunks' tails fall)
        host = eng.compute(rb)
        t_calls = []
        for _ in range(5):
            t = time.perf_counter()
            host = eng.compute(rb)
            t_calls.append(time.perf_counter() - t)
        t_host = min(t_calls)
        import numpy as np
        assert np.allclose(host, got, rtol=0, atol=1e-9)
        cells = sharding.region_cells(rb)
        row = {"workload": rshape, "regions": rb.n_regions, "reads": rb.n_reads, "pairs": rb.n_out, "cells": int(rb.cells()),
               "cells_per_region_min_median_max": [int(cells.min()), int(np.median(cells)), int(cells.max())],
               "gcups": round(r.plan.cells * 5 / el / 1e9, 1), "regions_per_s": round(rb.n_regions * 5 / el, 1),
               "ms_per_step": round(el / 5 * 1e3, 4), "launches_per_step": r.plan.num_launches,
               "dominant_kernel": r.plan.dominant_kernel,
               # what the launches sweep over what the metric counts: padding columns inside a lane group, empty haplotype slots
               "executed_per_cell": round(r.plan.executed_cells / max(r.plan.cells, 1), 4),
               "valu_f64": {"achieved": round(FLOP_PER_CELL * r.plan.cells / (sum(kms) / len(kms)) / 1e9, 3), "peak": VALU_F64_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(FLOP_PER_CELL * r.plan.cells / (sum(kms) / len(kms)) / 1e9 / VALU_F64_PEAK_TFLOPS, 4)},
               "host_buffers_incl_pcie": {"ms_per_call": round(t_host * 1e3, 3), "gcups": round(rb.cells() / t_host / 1e9, 1),
                                          "regions_per_s": round(rb.n_regions / t_host, 1),
                                          "ms_median_of_5": round(sorted(t_calls)[2] * 1e3, 3)},
               "oracle_sample": oracle_sample_diff(rb, got, 64, int(1.5e9))}
        r.close()
        return row

    def sw_measure(ref_off, ref, alt_off, alt, weights, strategy, cap, sample):
        """One phmm_sw_align call over host buffers, timed (three repeats), with the kernel's own HIP-event time, and the
        oracle (the reference's scalar arm in C) beside it on the first `sample` alignments."""
        import ctypes as C
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        from lorikeet_amd import _lib
        from oracle import oracle
        n = len(ref_off) - 1
        rl, al = np.diff(ref_off.astype(np.int64)), np.diff(alt_off.astype(np.int64))
        cells = int(np.sum(rl * al))
        cig_off = (np.arange(n + 1, dtype=np.uint64) * cap)
        cigar = np.zeros(n * cap, np.uint32)
        n_cig = np.zeros(n, np.uint32)
        off = np.zeros(n, np.int32)
        prm = _lib.SwParameters(*weights)
        pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
        args = (eng._h, n, pp(ref_off, _lib.u32p), pp(ref, _lib.u8p), pp(alt_off, _lib.u32p), pp(alt, _lib.u8p), C.byref(prm),
                strategy, pp(cig_off, _lib.u64p), pp(cigar, _lib.u32p), pp(n_cig, _lib.u32p), pp(off, C.POINTER(C.c_int32)))
        eng.set_switch("sw_clock", 1)                      # (block 0 of the aligner reports its shader clock: measurement runs only)
        assert eng.lib.phmm_sw_align(*args) == 0, eng.last_error()
        t = time.perf_counter()
        for _ in range(3):
            assert eng.lib.phmm_sw_align(*args) == 0
        dt = (time.perf_counter() - t) / 3
        kern_s = eng.stat("sw_kernel_us") / 1e6            # HIP events around the kernels of the last call
        second_default = eng.stat("sw_second_pass")        # alignments the full instance aligned again behind the tags-only sweep
        eng.set_switch("sw_lite", 0)                       # ... and the full instance alone (one pass)
        assert eng.lib.phmm_sw_align(*args) == 0
        t = time.perf_counter()
        assert eng.lib.phmm_sw_align(*args) == 0
        dt_full, full_s = time.perf_counter() - t, eng.stat("sw_kernel_us") / 1e6
        eng.set_switch("sw_lite", -1)
        assert eng.lib.phmm_sw_align(*args) == 0           # (the flag volume below is the default path's)
        flag_bytes = eng.stat("sw_backtrack_bytes")
        # the same call as ONE piece (no staging overlapped with the kernels: slower end to end, but the kernels run back to back
        # -- and in two passes where that pays: tags-only sweep, then the full instance over the alignments whose walk met a gap)
        eng.set_switch("sw_chunks", 1)
        one_s, one_again = 1e9, 0
        for _ in range(3):
            assert eng.lib.phmm_sw_align(*args) == 0
            if eng.stat("sw_kernel_us") / 1e6 < one_s:
                one_s, one_again = eng.stat("sw_kernel_us") / 1e6, eng.stat("sw_second_pass")
        eng.set_switch("sw_chunks", 0)
        k = min(n, sample)
        L = oracle.lib()
        cores = usable_cores()
        o_cig, o_n, o_off = np.zeros(k * cap, np.uint32), np.zeros(k, np.uint32), np.zeros(k, np.int32)
        oracle_strategy = {_lib.PHMM_SW_SOFTCLIP: 0, _lib.PHMM_SW_INDEL: 1, _lib.PHMM_SW_LEADING_INDEL: 2, _lib.PHMM_SW_IGNORE: 3}[strategy]

        def chunk(lo, hi):
            for a in range(lo, hi):
                tmp = np.zeros(int(rl[a] + al[a]) + 3, np.uint32)
                o = C.c_int32(0)
                m = L.oracle_sw_align(pp(ref[int(ref_off[a]):], _lib.u8p), int(rl[a]), pp(alt[int(alt_off[a]):], _lib.u8p), int(al[a]),
                                      *weights, oracle_strategy, pp(tmp, _lib.u32p), C.byref(o))
                o_n[a], o_off[a] = m, o.value
                o_cig[a * cap:a * cap + min(m, cap)] = tmp[:min(m, cap)]
        tc = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda r: chunk(*r), [(i, min(k, i + 64)) for i in range(0, k, 64)]))
        tc = time.perf_counter() - tc
        same = bool(np.array_equal(o_n, n_cig[:k]) and np.array_equal(o_off, off[:k]) and
                    all(np.array_equal(o_cig[a * cap:a * cap + min(int(o_n[a]), cap)], cigar[a * cap:a * cap + min(int(o_n[a]), cap)]) for a in range(k)))
        # what no implementation can avoid moving: both sequences in, the CIGAR and the offset out
        compulsory = int(rl.sum() + al.sum() + 4 * int(np.minimum(n_cig, cap).sum()) + 8 * n)
        return {"alignments": int(n), "cells": cells, "ms_per_call": round(dt * 1e3, 3),
                "alignments_per_s": round(n / dt, 1), "gcups_i32": round(cells / dt / 1e9, 1),
                "single_element_cigars": int(np.sum(n_cig == 1)), "cigar_elements_mean": round(float(n_cig.mean()), 2),
                "equal_to_oracle_on_sample": same, "sample": int(k),
                "kernel": {"ms": round(kern_s * 1e3, 3), "gcups_i32": round(cells / max(kern_s, 1e-9) / 1e9, 1),
                           "shader_clock_mhz": int(eng.stat("sw_clock_mhz")),
                           "second_pass_alignments": int(second_default),
                           "note": "the phmm_sw_align_kernel<L,K> launches of the last call (HIP events in the library, phmm_get_stat): "
                                   "a tags-only sweep, then the full instance over the alignments whose walk met a gap, where the "
                                   "handle expects few gaps; one pass otherwise"},
                "full_instance_only": {"ms_per_call": round(dt_full * 1e3, 3), "kernel_ms": round(full_s * 1e3, 3),
                                       "kernel_gcups_i32": round(cells / max(full_s, 1e-9) / 1e9, 1), "note": "switch sw_lite = 0"},
                "kernel_one_piece": {"ms": round(one_s * 1e3, 3), "gcups_i32": round(cells / max(one_s, 1e-9) / 1e9, 1),
                                     "second_pass_alignments": int(one_again),
                                     "note": "the same alignments as one piece (switch sw_chunks = 1): two passes where the handle expects few gaps"},
                "cpu_oracle": {"gcups_i32": round(int(np.sum(rl[:k] * al[:k])) / tc / 1e9, 3), "alignments_per_s": round(k / tc, 1),
                               "cores": cores, "kind": "port",
                               "note": "oracle/sw_oracle.c (the reference's scalar arm), ctypes calls from a thread pool"}}, \
            kern_s, flag_bytes, compulsory

    def smith_waterman():
        """every read of the batch's regions (up to 1 024) realigned to the first haplotype of its
        region (the shape of realign_reads_to_their_best_haplotype, src/assembly/assembly_based_caller_utils.rs:208-246:
        SoftClip, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS), host buffers, PCIe included."""
        import numpy as np
        from lorikeet_amd import _lib
        sub = batch.region_slice(0, min(1024, batch.n_regions))
        n = sub.n_reads
        alt_off, alt = sub.read_off, sub.read_bases
        reg_of_read = np.repeat(np.arange(sub.n_regions), np.diff(sub.region_read_off.astype(np.int64)))
        first_hap = sub.region_hap_off[:-1].astype(np.int64)[reg_of_read]
        h0, h1 = sub.hap_off.astype(np.int64)[first_hap], sub.hap_off.astype(np.int64)[first_hap + 1]
        ref_off = np.concatenate([[0], np.cumsum(h1 - h0)]).astype(np.uint32)
        ref = np.concatenate([sub.hap_bases[a:b] for a, b in zip(h0, h1)])
        row, kern_s, flag_bytes, compulsory = sw_measure(ref_off, ref, alt_off, alt, (10, -15, -30, -5), _lib.PHMM_SW_SOFTCLIP, 16, 4096)
        cells = row["cells"]
        pe, pwhy = pmc_entry("smith_waterman", int(n), "phmm_sw_align_kernel", "i32")
        row = {"call": "phmm_sw_align: reads -> first haplotype of their region, SoftClip, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS "
                       "(10,-15,-30,-5); host buffers, PCIe and CIGAR assembly included", **row}
        row["roofline"] = {"bound": "hbm", "achieved": round(compulsory / max(kern_s, 1e-9) / 1e9, 2), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(compulsory / max(kern_s, 1e-9) / 1e9 / HBM_PEAK_GBS, 5),
                           "algorithmic_bytes_per_launch": compulsory,
                           "traffic": pe["hbm_bytes_per_launch"] if pe else None,
                           "backtrack_flag_bytes_per_launch": int(flag_bytes),
                           "note": "algorithmic bytes = what any implementation must move: both sequences in, CIGAR elements and "
                                   "offset out.  The backtrack flags (4 bits per cell, written once, the cells on the backtrack "
                                   "path read once) are this implementation's choice and are what `traffic` mostly is; neither "
                                   "comes near HBM: the bound that binds is VALU issue, see valu_int32"
                                   + ("" if pe else "; traffic: " + pwhy)}
        # the cell is 16 instructions: 9 that issue every 2.55 clocks per wave64 (add, and, cndmask) and 7 that issue
        # every 4.3-4.8 (max, max3, alignbit, compare) -- tools/ubench/rates.hip; the cell body alone, no memory, no
        # branches, runs at 56-61 clocks per wave-level cell (tools/ubench/sw_cell.hip): that is `peak`
        if pe:
            rate = pe["valu_insts_per_launch"] / max(kern_s, 1e-9) / NUM_SIMD / 1e9
            row["valu_issue"] = {"achieved": round(rate, 4), "peak": VALU_ISSUE_PEAK_32, "unit": "G wave64-instr/s per SIMD",
                                 "frac": round(rate / VALU_ISSUE_PEAK_32, 4), "valu_per_cell": round(pe["valu_insts_per_launch"] * 64 / cells, 2),
                                 "mix_ceiling": SW_MIX_ISSUE_CEILING, "frac_of_mix_ceiling": round(rate / SW_MIX_ISSUE_CEILING, 4)}
        row["valu_int32"] = ({"valu_insts_per_cell": round(pe["valu_insts_per_launch"] * 64 / cells, 2),
                              "achieved": round(cells / max(kern_s, 1e-9) / 1e9, 1), "peak": SW_CELL_CEILING_GCUPS,
                              "unit": "GCUPS-i32", "frac": round(cells / max(kern_s, 1e-9) / 1e9 / SW_CELL_CEILING_GCUPS, 4),
                              "lane_ops_per_s": round(pe["valu_insts_per_launch"] * 64 / max(kern_s, 1e-9) / 1e12, 2),
                              "note": "(the default path sweeps with candidate tags only, 11 instructions per cell, and sends walks that met "
                                      "a gap through the full 16-instruction instance) peak = what the 16-instruction cell body alone sustains on the whole chip at four waves per "
                                      "SIMD (tools/ubench/sw_cell.hip, profiles/r02_ubench_sw_cell.txt: 61 clocks per wave-level "
                                      "cell at 2.4 GHz; nine of the instructions issue every 2.55 clocks, seven every 4.3-4.8: "
                                      "profiles/r02_ubench_rates.txt); valu_insts_per_cell = SQ_INSTS_VALU of the call "
                                      "(profiles/pmc_traffic.json, same kernel sources) x 64 lanes / cells"} if pe else None)
        row["indel_rich"] = optional(sw_indel_rich)
        return row

    def sw_indel_rich():
        """The same call on alignments that are NOT one match run: (a) 16 384 reads of 150 bases carrying 2-5 % indels and
        1 % mismatches against the 300-base window they come from (SoftClip, ALIGNMENT_TO_BEST_HAPLOTYPE parameters);
        (b) 2 048 haplotypes of ~400 bases with 1-3 indels of 1-12 bases and a few SNPs against their 400-base reference
        (InDel, NEW_SW_PARAMETERS 200,-150,-260,-11: the shape of calculate_cigar, reads/cigar_builder + haplotype.rs)."""
        import numpy as np
        from lorikeet_amd import _lib
        rng = np.random.default_rng(77)
        acgt = np.frombuffer(b"ACGT", np.uint8)

        def mutate(src, indel_rate, snp_rate):
            out, i = [], 0
            while i < len(src):
                u = rng.random()
                if u < indel_rate / 2:
                    out.append(acgt[rng.integers(0, 4, int(rng.integers(1, 4)))])     # insertion of 1-3 bases
                elif u < indel_rate:
                    i += int(rng.integers(1, 4))                                       # deletion of 1-3 bases
                    continue
                out.append(src[i:i + 1] if rng.random() >= snp_rate else acgt[rng.integers(0, 4, 1)])
                i += 1
            return np.concatenate(out) if out else src[:1]

        def pairs(n, ref_len, make_alt):
            refs = [acgt[rng.integers(0, 4, ref_len)] for _ in range(n)]
            alts = [make_alt(r) for r in refs]
            ro = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.uint32)
            ao = np.concatenate([[0], np.cumsum([len(a) for a in alts])]).astype(np.uint32)
            return ro, np.concatenate(refs), ao, np.concatenate(alts)

        def read_of(r):
            s = int(rng.integers(0, len(r) - 160))
            return mutate(r[s:s + 150], float(rng.uniform(0.02, 0.05)), 0.01)

        def hap_of(r):
            h = r.copy()
            for _ in range(int(rng.integers(1, 4))):
                at, ln = int(rng.integers(20, len(h) - 40)), int(rng.integers(1, 13))
                h = np.concatenate([h[:at], acgt[rng.integers(0, 4, ln)], h[at:]]) if rng.random() < 0.5 else np.concatenate([h[:at], h[at + ln:]])
            snp = rng.integers(0, len(h), 3)
            h[snp] = acgt[rng.integers(0, 4, 3)]
            return h
        out = {}
        for name, (data, weights, strategy, cap) in {
                "reads_150_with_2_to_5_percent_indels": (pairs(16384, 300, read_of), (10, -15, -30, -5), _lib.PHMM_SW_SOFTCLIP, 48),
                "haplotype_to_reference_400x400": (pairs(2048, 400, hap_of), (200, -150, -260, -11), _lib.PHMM_SW_INDEL, 32)}.items():
            row, kern_s, flag_bytes, compulsory = sw_measure(*data, weights, strategy, cap, 2048)
            row["weights"] = list(weights)
            out[name] = row
        return out

    def realign(likelihoods):
        """The step behind the likelihoods (assembly_based_caller_utils.rs:208-246): best allele per read with the
        reference's tie-breaking priorities, and every read aligned to its best haplotype, in one call
        (phmm_realign_to_best: the haplotypes cross the bus once, the index never leaves the device); host buffers."""
        import numpy as np
        from lorikeet_amd import realign as rl
        from oracle import oracle
        sub = batch.region_slice(0, min(1024, batch.n_regions))
        lk = np.ascontiguousarray(likelihoods[:int(sub.out_off[-1])])
        rng = np.random.default_rng(11)
        # the first haplotype of a region is its reference, CIGARs of one to three elements (what assembled haplotypes have)
        is_ref = np.zeros(sub.n_haps, bool)
        is_ref[sub.region_hap_off[:-1]] = True
        pri = rl.haplotype_alignment_tiebreaking_priority(is_ref, np.where(is_ref, 1, rng.integers(1, 4, sub.n_haps)))
        best, res = rl.realign_reads_to_their_best_haplotype(eng, sub, lk, pri)
        t = time.perf_counter()
        for _ in range(3):
            best, res = rl.realign_reads_to_their_best_haplotype(eng, sub, lk, pri)
        dt_py = (time.perf_counter() - t) / 3
        # the C call alone (the Python mirror builds one result object per read)
        import ctypes as C
        from lorikeet_amd import _lib
        n = sub.n_reads
        cap = 16
        cig_off = np.arange(n + 1, dtype=np.uint64) * cap
        cigar, n_cig, off = np.zeros(n * cap, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
        b_idx, b_lk, b_conf = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
        prm = _lib.SwParameters(10, -15, -30, -5)
        pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
        i32p = C.POINTER(C.c_int32)
        args = (eng._h, sub.n_regions, pp(sub.region_read_off, _lib.u32p), pp(sub.region_hap_off, _lib.u32p), pp(sub.read_off, _lib.u32p),
                pp(sub.read_bases, _lib.u8p), pp(sub.hap_off, _lib.u32p), pp(sub.hap_bases, _lib.u8p), pp(sub.out_off, _lib.u64p), pp(lk, _lib.f64p),
                None, pp(pri, i32p), 0.2, C.byref(prm), _lib.PHMM_SW_SOFTCLIP, pp(cig_off, _lib.u64p), pp(cigar, _lib.u32p), pp(n_cig, _lib.u32p),
                pp(off, i32p), pp(b_idx, i32p), pp(b_lk, _lib.f64p), pp(b_conf, _lib.f64p))
        assert eng.lib.phmm_realign_to_best(*args) == 0, eng.last_error()
        t = time.perf_counter()
        for _ in range(5):
            assert eng.lib.phmm_realign_to_best(*args) == 0
        dt = (time.perf_counter() - t) / 5
        kern_s = eng.stat("sw_kernel_us") / 1e6
        # oracle on a sample of regions: best alleles equal, CIGARs and offsets equal
        same, checked = True, 0
        nr = np.diff(sub.region_read_off.astype(np.int64))
        nh = np.diff(sub.region_hap_off.astype(np.int64))
        for g in range(0, sub.n_regions, max(1, sub.n_regions // 8)):
            m = lk[int(sub.out_off[g]):int(sub.out_off[g]) + nr[g] * nh[g]].reshape(nr[g], nh[g])
            wb, wl, wc = oracle.best_alleles(m.T, pri[sub.region_hap_off[g]:sub.region_hap_off[g + 1]], 0.2)
            r0 = int(sub.region_read_off[g])
            same = same and np.array_equal(wb, b_idx[r0:r0 + nr[g]]) and np.array_equal(wl, b_lk[r0:r0 + nr[g]]) and np.array_equal(wc, b_conf[r0:r0 + nr[g]])
            for r in range(r0, r0 + min(int(nr[g]), 32)):
                hp = int(sub.region_hap_off[g]) + int(wb[r - r0])
                cg, of = oracle.sw_align(sub.hap_bases[int(sub.hap_off[hp]):int(sub.hap_off[hp + 1])],
                                         sub.read_bases[int(sub.read_off[r]):int(sub.read_off[r + 1])], [10, -15, -30, -5], "SoftClip")
                same = same and of == off[r] and np.array_equal(cg, cigar[r * cap:r * cap + int(n_cig[r])])
                checked += 1
        cells = int(np.sum(np.diff(sub.read_off.astype(np.int64)) * np.diff(sub.hap_off.astype(np.int64))[sub.region_hap_off[:-1].astype(np.int64)[np.repeat(np.arange(sub.n_regions), nr)] + b_idx]))
        # ... and the projection of those alignments onto the reference (the rest of create_read_aligned_to_ref): the
        # synthetic haplotypes differ from their region's first one by SNVs, so their CIGARs are one M element
        hlen = np.diff(sub.hap_off.astype(np.int64))
        hc_off = np.arange(sub.n_haps + 1, dtype=np.uint32)
        hc = ((hlen << 4) | 0).astype(np.uint32)
        hs = np.zeros(sub.n_haps, np.uint32)
        rrh = np.zeros(sub.n_regions, np.int32)
        rstart = (1000 + 1000 * np.arange(sub.n_regions)).astype(np.uint64)
        rlen = np.diff(sub.read_off.astype(np.int64))
        oc_off = np.arange(n + 1, dtype=np.uint32)
        oc = ((rlen << 4) | 0).astype(np.uint32)
        out_off = np.arange(n + 1, dtype=np.uint64) * 8
        out_c, n_out, pos, status = np.zeros(n * 8, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64), np.zeros(n, np.int32)
        pargs = (eng._h, sub.n_regions, pp(sub.region_read_off, _lib.u32p), pp(sub.region_hap_off, _lib.u32p), pp(sub.read_off, _lib.u32p),
                 pp(sub.read_bases, _lib.u8p), pp(sub.hap_off, _lib.u32p), pp(sub.hap_bases, _lib.u8p), pp(rrh, i32p), pp(rstart, _lib.u64p),
                 pp(hc_off, _lib.u32p), pp(hc, _lib.u32p), pp(hs, _lib.u32p), pp(b_idx, i32p), pp(cig_off, _lib.u64p), pp(cigar, _lib.u32p),
                 pp(n_cig, _lib.u32p), pp(off, i32p), pp(oc_off, _lib.u32p), pp(oc, _lib.u32p), pp(out_off, _lib.u64p), pp(out_c, _lib.u32p),
                 pp(n_out, _lib.u32p), pp(pos, C.POINTER(C.c_int64)), pp(status, i32p))
        assert eng.lib.phmm_project_to_reference(*pargs) == 0, eng.last_error()
        t = time.perf_counter()
        for _ in range(5):
            assert eng.lib.phmm_project_to_reference(*pargs) == 0
        dt_proj = (time.perf_counter() - t) / 5
        # ... and all three steps in one call: nothing but the results comes back
        b2, l2, c2 = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
        out2, n_out2, pos2, status2 = np.zeros(n * 8, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64), np.zeros(n, np.int32)
        fargs = (eng._h, sub.n_regions, pp(sub.region_read_off, _lib.u32p), pp(sub.region_hap_off, _lib.u32p), pp(sub.read_off, _lib.u32p),
                 pp(sub.read_bases, _lib.u8p), pp(sub.hap_off, _lib.u32p), pp(sub.hap_bases, _lib.u8p), pp(sub.out_off, _lib.u64p), pp(lk, _lib.f64p),
                 None, pp(pri, i32p), 0.2, C.byref(prm), _lib.PHMM_SW_SOFTCLIP, pp(rrh, i32p), pp(rstart, _lib.u64p), pp(hc_off, _lib.u32p),
                 pp(hc, _lib.u32p), pp(hs, _lib.u32p), pp(oc_off, _lib.u32p), pp(oc, _lib.u32p), pp(out_off, _lib.u64p), pp(out2, _lib.u32p),
                 pp(n_out2, _lib.u32p), pp(pos2, C.POINTER(C.c_int64)), pp(status2, i32p), pp(b2, i32p), pp(l2, _lib.f64p), pp(c2, _lib.f64p))
        assert eng.lib.phmm_realign_reads(*fargs) == 0, eng.last_error()
        t = time.perf_counter()
        for _ in range(5):
            assert eng.lib.phmm_realign_reads(*fargs) == 0
        dt_fused = (time.perf_counter() - t) / 5
        used = np.arange(8)[None, :] < n_out[:, None]  # (what lies behind a CIGAR in its slot is not defined)
        fused_same = bool(np.array_equal(b2, b_idx) and np.array_equal(status2, status) and np.array_equal(pos2, pos) and
                          np.array_equal(n_out2, n_out) and np.array_equal(out2.reshape(n, 8)[used], out_c.reshape(n, 8)[used]))
        proj_same, proj_checked = True, 0
        for g in range(0, sub.n_regions, max(1, sub.n_regions // 8)):
            r0 = int(sub.region_read_off[g])
            h0 = int(sub.region_hap_off[g])
            ref = sub.hap_bases[int(sub.hap_off[h0]):int(sub.hap_off[h0 + 1])]
            for r in range(r0, r0 + min(int(nr[g]), 32)):
                hp = h0 + int(b_idx[r])
                want = oracle.create_read_aligned_to_ref(cigar[r * cap:r * cap + int(n_cig[r])], int(off[r]), hc[hp:hp + 1], 0, int(rstart[g]), ref,
                                                         sub.read_bases[int(sub.read_off[r]):int(sub.read_off[r + 1])], oc[r:r + 1])
                proj_same = proj_same and status[r] == 0 and want == (int(pos[r]), oracle.cigar_to_string(out_c[r * 8:r * 8 + int(n_out[r])]))
                proj_checked += 1
        return {"call": "phmm_realign_to_best: best allele per read (haplotype_alignment_tiebreaking_priority) + SoftClip alignment of the read "
                        "to it, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS; host buffers, PCIe included",
                "reads": int(n), "regions": int(sub.n_regions), "ms_per_call": round(dt * 1e3, 3), "reads_per_s": round(n / dt, 1),
                "gcups_i32": round(cells / dt / 1e9, 1), "sw_kernels_ms": round(kern_s * 1e3, 3),
                "informative_reads": int(np.sum(b_conf > 0.2)), "python_mirror_ms_per_call": round(dt_py * 1e3, 1),
                "equal_to_oracle_on_sample": bool(same), "sample_alignments": checked,
                "project_to_reference": {"call": "phmm_project_to_reference on the same reads: the alignments projected through the haplotypes' "
                                                 "CIGARs, left-aligned, clips restored (the rest of create_read_aligned_to_ref); host buffers",
                                         "ms_per_call": round(dt_proj * 1e3, 3), "reads_per_s": round(n / dt_proj, 1),
                                         "realigned": int(np.sum(status == 0)), "with_indels": int(np.sum(n_out > 1)),
                                         "equal_to_oracle_on_sample": bool(proj_same), "sample_reads": proj_checked},
                "realign_reads_one_call": {"call": "phmm_realign_reads: best alleles + alignments + projection, the alignments never leave the device",
                                           "ms_per_call": round(dt_fused * 1e3, 3), "reads_per_s": round(n / dt_fused, 1),
                                           "equal_to_the_separate_calls": fused_same}}

    class Dist1:  # rank-0-only rows: same timing code, no cross-rank barrier
        def __init__(self, d):
            self.torch, self.dev = d.torch, d.dev

        def barrier(self):
            self.torch.cuda.synchronize(self.dev)

        def max(self, x):
            return float(x)

    single = f32_row = engine_row = calls_row = ragged_row = sw_row = None
    # (BENCH_ROWS=ragged,sw ... : only these secondary rows -- developer runs; the driver's run has them all)
    only_rows = set(filter(None, os.environ.get("BENCH_ROWS", "").split(",")))
    want = lambda name: not only_rows or name in only_rows  # noqa: E731
    if rank == 0 and extras:
        if want("single"):
            single = optional(single_region)
        if not a.f32_first and want("f32"):
            f32_row = optional(f32_first)
        if want("engine"):
            engine_row = optional(engine_call)
        if world == 1 and want("calls"):
            calls_row = optional(host_calls)
        if a.workload == "config2":
            if want("ragged"):
                ragged_row = optional(ragged)
            if want("sw"):
                sw_row = optional(smith_waterman)

    if rank == 0:
        got = out.cpu().numpy()
        assert (got <= 0).all(), "non-finite or positive likelihoods"
        # one phmm_batch_launch = every kernel of the plan (one for a uniform batch); the roofline is taken over all of it
        mean_kernel_s = sum(kern_ms) / len(kern_ms) / 1e3
        alg_bytes = plan.algorithmic_bytes
        achieved_gbs = alg_bytes / mean_kernel_s / 1e9
        e, why = pmc_entry(a.workload, regions, plan.dominant_kernel, "f32_first" if a.f32_first else "f64")
        line = {
            "metric": "PairHMM cell-updates/s (GCUPS)",
            "value": round(cells_total * a.steps / elapsed / 1e9, 2),
            "unit": "GCUPS",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            # (BENCH_DIST_BACKEND=gloo on a box with fewer devices than ranks only exercises the N > 1 code path: ranks share a
            # device, and the line says so -- it is not a scaling measurement)
            "devices_aliased": bool(world > max(torch.cuda.device_count(), 1)),
            "dtype": "f32 first, f64 redo (PMC pass only)" if a.f32_first else "f64", "data": "synthetic",
            "config": {"workload": "config3: the same %d regions (seed %d) at every N, sharded over the %d rank(s) in contiguous "
                                   "cell-balanced ranges (BASELINE.json configs[2]/[3]): %s" % (strong["regions"], strong["seed"], world, shape)
                       if strong else
                       "%s x %d regions per GPU: %s (BASELINE.json configs[1] shape, batched per "
                       "SURVEY 8d)" % (a.workload, regions, shape)
                       if a.workload == "config2" else "%s x %d regions per GPU: %s" % (a.workload, regions, shape),
                       "per_rank_cells": per_rank_cells,
                       "regions_per_gpu": regions, "pairs_per_gpu": int(batch.n_out),
                       "cells_per_gpu_per_step": int(plan.cells), "seed": a.seed,
                       "sharding": "regions, one process per GPU, no collective",
                       # (what this process changed in its own environment; the threads_bench children run with the library's defaults)
                       "env": None if own_queue_given else "PHMM_REGION_OWN_QUEUE=0 in this process only"},
            "regions_per_s": round(regions_total * a.steps / elapsed, 1),
            # The bound that BINDS leads: FP64 vector flops of the reference recurrence against the chip's FP64
            # VALU peak.  The HBM figure the metric's wording asks for stays beside it under "hbm" -- 2.3e-3 compulsory bytes per
            # cell make it a fraction of a per cent whatever the kernel does; `traffic` is the PMC byte count against it.
            # (--f32-first, the PMC passes of the opt-in mode: its sweep computes in f32, so its bound is the FP32 vector peak)
            "roofline": {"bound": "valu_f32" if a.f32_first else "valu_f64", "achieved": round(FLOP_PER_CELL * plan.cells / mean_kernel_s / 1e12, 3),
                         "peak": VALU_F32_PEAK_TFLOPS if a.f32_first else VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(FLOP_PER_CELL * plan.cells / mean_kernel_s / 1e12 / (VALU_F32_PEAK_TFLOPS if a.f32_first else VALU_F64_PEAK_TFLOPS), 4),
                         "traffic": e["hbm_bytes_per_launch"] if e else None, "l2_hit_rate": e.get("l2_hit_rate") if e else None,
                         "hbm": {"bound": "hbm", "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(achieved_gbs / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": int(alg_bytes)},
                         "kernel": plan.dominant_kernel, "kernel_ms": round(mean_kernel_s * 1e3, 4),
                         "kernels_per_launch": plan.num_launches,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "src_hash": source_hash(),
                         "kernel_rocprof": e.get("kernel") if e else None,
                         "note": "achieved = 12 flop per cell (pair_hmm.rs:561-587) x cells per launch / the launch's mean duration, measured "
                                 "in THIS run (HIP events on the launch stream); hbm.achieved = algorithmic bytes per launch (SURVEY 8d) over "
                                 "the same duration; traffic, l2_hit_rate and valu_issue come from the committed rocprofv3 --pmc passes of "
                                 "the same command (profiles/pmc_traffic.json), used only when they were taken on this kernel built from "
                                 "these sources (src_hash)" + ("" if e else "; traffic: " + why)},
            "valu_f64": None if a.f32_first else {"achieved": round(FLOP_PER_CELL * plan.cells / mean_kernel_s / 1e12, 3),
                         "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(FLOP_PER_CELL * plan.cells / mean_kernel_s / 1e12 / VALU_F64_PEAK_TFLOPS, 4),
                         "flop_per_cell": FLOP_PER_CELL,
                         "note": "flop_per_cell counts the reference recurrence (pair_hmm.rs:561-587); the kernel "
                                 "executes 10 (4 FMA + 2 MUL) after folding three factors into the row constants"},
            "oracle_sample": optional(lambda: oracle_sample_diff(batch, got, 2, int(2e8))) if extras else None,
        }
        if e and e.get("valu_insts_per_launch"):  # the bound that actually binds: wave64 VALU issue slots (every VALU op costs one, FP64 or not)
            valu_insts = e["valu_insts_per_launch"]
            rate = valu_insts / mean_kernel_s / NUM_SIMD / 1e9
            line["valu_issue"] = {
                "achieved": round(rate, 4), "peak": VALU_ISSUE_PEAK, "unit": "G wave64-instr/s per SIMD",
                "frac": round(rate / VALU_ISSUE_PEAK, 4), "valu_per_cell": round(valu_insts * 64 / plan.cells, 3),
                "same_mix_ubench": VALU_ISSUE_UBENCH, "source": e.get("source"),
                "note": "peak = 2.4 GHz / 4 clk; same_mix_ubench = the 7-instruction cell body alone (hoisted v_cmp -> SGPR "
                        "mask form) at 2 waves/SIMD on the whole chip (tools/ubench/issue.hip: 3.9 clk per instruction at "
                        "the 2.3 GHz the chip sustains)"}
        else:
            line["valu_issue"] = {"achieved": None, "note": why}
        line["config3_10k"] = config3_row
        line["config5_256"] = config5_row
        line["ragged"] = ragged_row
        line["single_region"] = single
        line["f32_first"] = f32_row
        line["engine_call"] = engine_row
        line["smith_waterman"] = sw_row
        if a.workload == "config2" and extras:
            line["realign_to_best"] = optional(lambda: realign(got))
        if calls_row is not None:
            line["host_calls"] = calls_row
        if world == 1 and not a.no_cpu_baseline and extras:
            scalar, simd = cpu_baselines(batch)
            line["cpu_baseline"] = scalar
            line["cpu_baseline_simd"] = simd
        full_path = write_full_record(line)
        print(json.dumps(compact_line(line, full_path), separators=(",", ":")), flush=True)
    res.close()
    D.close()


if __name__ == "__main__":
    main()
