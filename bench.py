#!/usr/bin/env python
"""bench.py -- PairHMM cell-updates/s (GCUPS) on MI355X, the metric BASELINE.json names.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (phmm_batch_launch: the forward kernels, nothing else) over
one batch of synthetic assembly regions already resident in HBM.  Workload at every N: per GPU
`--regions` (default 1024) regions of the BASELINE.json configs[1] shape -- 128 reads x 8
haplotypes, 150 bp reads, 300 bp haplotypes -- i.e. SURVEY.md 8(d)'s batched form of config 2;
rank r draws its own regions (seed base+r): regions shard across GPUs with no collective ("weak"
scaling).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     algorithmic HBM bytes (5*sum R + sum H + 8*Nr*Nh per region) per launch
               / the dominant kernel's mean launch duration (HIP events on the launch stream), vs
               8 TB/s.  `traffic` = HBM bytes per launch from rocprofv3 PMC passes, read from
               profiles/ (null if no summary for this workload is committed).  The path is NOT
               HBM-bound: `valu_f64` carries the binding roofline next to it.
  cpu_baseline the CPU oracle (C port of the reference's scalar path, oracle/) on the host cores of
               this box, rank 0 at N=1 only, on a bounded sample of the same regions.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_F64_PEAK_TFLOPS = 78.6  # vector FP64 peak (256 CU * 2.4 GHz * 128 flop/clk)
NUM_SIMD = 1024              # 256 CUs x 4
VALU_ISSUE_PEAK = 0.6        # G wave64 VALU instructions / s / SIMD at 2.4 GHz, one per 4 clk
VALU_ISSUE_UBENCH = 0.595    # what the kernel's own cell body sustains alone (tools/ubench/issue.hip, 2 waves/SIMD)
FLOP_PER_CELL = 12           # M 4 mul + 2 add, I 2+1, D 2+1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=1024, help="regions per GPU per step")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5"])
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flush-caches", action="store_true",
                    help="PMC calibration only: overwrite a 1 GiB buffer before every launch so that no input byte "
                         "survives in L2 / Infinity Cache from the previous launch (the timing then includes the fill)")
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed loop (no single_region / engine_call / cpu_baseline rows): for PMC passes")
    return ap.parse_args()


def make_workload(name, n_regions, seed):
    from lorikeet_amd import synthetic
    if name == "config2":
        return synthetic.config2(n_regions, seed=seed), "128 reads x 8 haps, R=150, H=300"
    if name == "config3":
        return synthetic.config3(n_regions, seed=seed), "128 reads x 8 haps, H=300, R in {100,150,250}"
    return synthetic.config5(n_regions, seed=seed), "512 reads x 64 haps, R=150, H=400"


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


def cpu_baseline(batch, budget_s=20.0):
    """Oracle ("port" of the reference's scalar path) on all usable host cores, bounded sample."""
    from oracle import oracle
    cores = usable_cores()
    try:
        oracle.build(native=True)
        native = True
    except Exception:
        native = False
    # calibrate on one region per core, then size the sample to ~budget_s
    n0 = min(batch.n_regions, cores)
    sub = batch.region_slice(0, n0)
    t = time.perf_counter()
    oracle.compute_batch(sub.as_dict(), n_threads=cores, native=native)
    dt0 = time.perf_counter() - t
    rounds = max(1, min(int(budget_s / max(dt0, 1e-3)), batch.n_regions // n0))
    n1 = n0 * rounds
    sub = batch.region_slice(0, n1)
    t = time.perf_counter()
    oracle.compute_batch(sub.as_dict(), n_threads=cores, native=native)
    dt = time.perf_counter() - t
    return {"value": round(sub.cells() / dt / 1e9, 4), "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": "first %d regions of rank 0's batch (%.3g cells, %.1f s); oracle/pairhmm_oracle.c, f64 scalar, "
                      "one region per pthread task%s; cores = min(affinity, cgroup cpu quota) of %d logical CPUs"
                      % (n1, sub.cells(), dt, ", -march=native" if native else "", os.cpu_count() or 1)}


def pmc_traffic(workload, regions):
    """(HBM bytes per launch, L2 hit rate, wave64 VALU instructions per launch) measured with rocprofv3 PMC
    passes of this very command (profiles/pmc_traffic.json; tools/profile.sh + tools/rocpd_summary.py)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        for e in json.load(open(path)):
            if e["workload"] == workload and e["regions"] == regions:
                return e["hbm_bytes_per_launch"], e.get("l2_hit_rate"), e.get("valu_insts_per_launch")
    except Exception:
        pass
    return None, None, None


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world))
    # one process per GPU; BENCH_DIST_BACKEND=gloo (ranks may then share a device) exists only to exercise the
    # N>1 code path on a single-GPU box
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:  # barrier + max-over-ranks only; no data-path collective
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from lorikeet_amd import HipPairHMMEngine
    batch, shape = make_workload(a.workload, a.regions, a.seed + rank)
    eng = HipPairHMMEngine(dev_index)
    plan = eng.plan(batch)
    tens = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in
            ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    out = torch.empty(batch.n_out, dtype=torch.float64, device=dev)
    plan.bind_torch(tens, out)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize(dev)

    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev) if a.flush_caches else None
    with torch.cuda.stream(stream):
        for _ in range(a.warmup):
            plan.launch(sh)
        barrier()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        t0 = time.perf_counter()
        ev[0].record(stream)
        for i in range(a.steps):
            if flush is not None:
                flush.fill_(float(i))
            plan.launch(sh)
            ev[i + 1].record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
    plan.status()  # raises if any likelihood came out > 0
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)]

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    cells_total = plan.cells * world  # identical shapes on every rank
    regions_total = batch.n_regions * world

    single = None
    if rank == 0 and not a.main_only:  # configs[1] literally: ONE region per launch (latency mode: the planner spreads it over all SIMDs)
        try:
            one, _ = make_workload(a.workload, 1, a.seed + 7919)
            p1 = eng.plan(one)
            t1 = {k: torch.from_numpy(getattr(one, k)).to(dev) for k in
                  ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
            o1 = torch.empty(one.n_out, dtype=torch.float64, device=dev)
            p1.bind_torch(t1, o1)
            with torch.cuda.stream(stream):
                for _ in range(10):
                    p1.launch(sh)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(100):
                    p1.launch(sh)
                e1.record(stream)
                stream.synchronize()
            us = e0.elapsed_time(e1) * 10.0
            single = {"regions": 1, "us_per_region": round(us, 2), "gcups": round(p1.cells / us / 1e3, 1),
                      "kernel": p1.dominant_kernel, "note": "one region per launch, back-to-back launches on one stream"}
            p1.close()
        except Exception as exc:  # an optional row must never cost the bench line
            single = {"error": repr(exc)}

    f32_row = None
    if rank == 0 and not a.main_only:  # the opt-in PHMM_FLAG_F32_FIRST mode on the same resident batch (never `value`)
        try:
            e32 = HipPairHMMEngine(dev_index, f32_first=True)
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
            f32_row = {"value": round(p32.cells / ms32 / 1e6, 2), "unit": "GCUPS", "ms_per_step": round(ms32, 4),
                       "kernel": p32.dominant_kernel, "dtype": "f32 first, f64 redo of what f32 cannot be trusted with",
                       "max_abs_diff_vs_f64": float((out32 - out).abs().max().item()), "tolerance": 1e-5,
                       "note": "opt-in flag of phmm_create, what the reference's vector arm (gkl) does; default and `value` are f64"}
            p32.close()
            e32.close()
        except Exception as exc:  # an optional row must never cost the bench line
            f32_row = {"error": repr(exc)}

    engine_row = None
    if rank == 0 and not a.main_only:  # SURVEY 8(f1/f2): the engine-level call (pre-step + PairHMM + normalise/disqualify), host buffers
        try:
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
            engine_row = {"call": "phmm_engine_compute (PCR model conservative, dynamic disqualification), host buffers, "
                                  "PCIe included", "regions": nreg, "ms_per_call": round(te * 1e3, 3),
                          "gcups_incl_pcie": round(sub.cells() / te / 1e9, 1), "reads_kept_fraction": round(float(keep.mean()), 4)}
        except Exception as exc:  # an optional row must never cost the bench line
            engine_row = {"error": repr(exc)}

    calls_row = None
    if rank == 0 and world == 1 and not a.main_only:
        # The reference's call granularity (one region per compute_likelihoods call from every rayon worker), host buffers,
        # PCIe included: tools/threads_bench (C++ threads on the C ABI, built with the library) in three configurations.
        try:
            import re
            import subprocess
            exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "threads_bench")

            def point(mode, threads, per_call):
                env = dict(os.environ, TB_MODE=mode, TB_THREADS=str(threads))
                r = subprocess.run([exe, "1.0", "128", "8", "150", "300", str(per_call)], env=env, capture_output=True,
                                   text=True, timeout=60)
                m = re.search(r"threads:\s+(\d+) regions/s\s+([\d.]+) GCUPS", r.stdout)
                return {"regions_per_s": int(m.group(1)), "gcups_incl_pcie": float(m.group(2))}
            calls_row = {
                "note": "config-2 regions through host buffers (PCIe, planning and staging included), C++ caller threads",
                "one_region_per_call_8_threads_own_handles": point("own", 8, 1),
                "one_region_per_call_32_threads_shared_handle_submit_wait": point("shared", 32, 1),
                "eight_regions_per_call_4_threads_own_handles": point("own", 4, 8)}
        except Exception as exc:  # an optional row must never cost the bench line
            calls_row = {"error": repr(exc)}

    if rank == 0:
        res = out.cpu().numpy()
        assert (res <= 0).all(), "non-finite or positive likelihoods"
        mean_kernel_s = sum(kern_ms) / len(kern_ms) / 1e3 / max(plan.num_launches, 1)
        alg_bytes = plan.algorithmic_bytes
        achieved_gbs = alg_bytes / mean_kernel_s / 1e9
        line = {
            "metric": "PairHMM cell-updates/s (GCUPS)",
            "value": round(cells_total * a.steps / elapsed / 1e9, 2),
            "unit": "GCUPS",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s x %d regions per GPU: %s (BASELINE.json configs[1] shape, batched per "
                                   "SURVEY 8d)" % (a.workload, a.regions, shape)
                       if a.workload == "config2" else "%s x %d regions per GPU: %s" % (a.workload, a.regions, shape),
                       "regions_per_gpu": a.regions, "pairs_per_gpu": int(batch.n_out),
                       "cells_per_gpu_per_step": int(plan.cells), "seed": a.seed,
                       "sharding": "regions, one process per GPU, no collective"},
            "regions_per_s": round(regions_total * a.steps / elapsed, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(a.workload, a.regions)[0],
                         "l2_hit_rate": pmc_traffic(a.workload, a.regions)[1],
                         "kernel": plan.dominant_kernel, "kernel_ms": round(mean_kernel_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "note": "compulsory traffic is 2.3e-3 B/cell: the path is FP64-VALU bound, see valu_f64"},
            "valu_f64": {"achieved": round(FLOP_PER_CELL * plan.cells / mean_kernel_s / 1e12, 3),
                         "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(FLOP_PER_CELL * plan.cells / mean_kernel_s / 1e12 / VALU_F64_PEAK_TFLOPS, 4),
                         "flop_per_cell": FLOP_PER_CELL,
                         "note": "flop_per_cell counts the reference recurrence (pair_hmm.rs:561-587); the kernel "
                                 "executes 10 (4 FMA + 2 MUL) after folding three factors into the row constants"},
        }
        valu_insts = pmc_traffic(a.workload, a.regions)[2]
        if valu_insts:  # the bound that actually binds: wave64 VALU issue slots (every VALU op costs one, FP64 or not)
            rate = valu_insts / mean_kernel_s / NUM_SIMD / 1e9
            line["valu_issue"] = {
                "achieved": round(rate, 4), "peak": VALU_ISSUE_PEAK, "unit": "G wave64-instr/s per SIMD",
                "frac": round(rate / VALU_ISSUE_PEAK, 4), "valu_per_cell": round(valu_insts * 64 / plan.cells, 3),
                "same_mix_ubench": VALU_ISSUE_UBENCH,
                "note": "peak = 2.4 GHz / 4 clk; same_mix_ubench = the 7-instruction cell body alone (hoisted v_cmp -> SGPR "
                        "mask form) at 2 waves/SIMD on the whole chip (tools/ubench/issue.hip: 3.9 clk per instruction at "
                        "the 2.3 GHz the chip sustains)"}
        line["single_region"] = single
        line["f32_first"] = f32_row
        line["engine_call"] = engine_row
        if calls_row is not None:
            line["host_calls"] = calls_row
        if world == 1 and not a.no_cpu_baseline and not a.main_only:
            line["cpu_baseline"] = cpu_baseline(batch)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
