"""Developer tool: soak parity run -- random batches of widely varying shape through the C ABI vs the oracle
(all host cores), for a wall-clock budget.  usage: python tools/soak.py [seconds] [seed]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lorikeet_amd import HipPairHMMEngine
from lorikeet_amd.batch import Read, RegionBatch
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
alpha = np.frombuffer(b"ACGT", np.uint8)
alphan = np.frombuffer(b"ACGTN", np.uint8)
from bench import usable_cores
cores = usable_cores()  # affinity capped by the cgroup CPU quota: more oracle threads only oversubscribe
eng = HipPairHMMEngine(0)
eng32 = HipPairHMMEngine(0, f32_first=True)  # opt-in mode: same batches, gate 1e-5 (north_star) instead of 1e-9
shared = HipPairHMMEngine(0)  # phmm_submit / phmm_wait from several threads
worst32 = 0.0
t_end = time.time() + budget
worst, n_batches, n_pairs, n_cells = 0.0, 0, 0, 0
kinds = {}
while time.time() < t_end:
    kind = rng.choice(["typical", "typical", "typical", "tiny", "longhap", "longread", "manyhaps", "withN", "lowq", "wideq"])
    regions = []
    n_regions = int(rng.integers(1, 40)) if kind in ("typical", "tiny", "withN", "lowq", "wideq") else int(rng.integers(1, 4))
    for _ in range(n_regions):
        if kind == "tiny":
            nr, nh, hl, rl = int(rng.integers(0, 5)), int(rng.integers(1, 4)), (1, 40), (0, 30)
        elif kind == "longhap":
            nr, nh, hl, rl = int(rng.integers(1, 6)), int(rng.integers(1, 4)), (1500, 2600), (50, 200)
        elif kind == "longread":
            nr, nh, hl, rl = int(rng.integers(1, 4)), int(rng.integers(1, 3)), (100, 700), (800, 3300)
        elif kind == "manyhaps":
            nr, nh, hl, rl = int(rng.integers(4, 30)), int(rng.integers(20, 70)), (200, 420), (80, 160)
        else:
            nr, nh, hl, rl = int(rng.integers(1, 40)), int(rng.integers(1, 10)), (60, 450), (20, 260)
        a = alphan if kind == "withN" else alpha
        root = a[rng.integers(0, len(a), int(rng.integers(*hl)))]
        haps = []
        for j in range(nh):
            h = root.copy()
            if j:
                for _ in range(int(rng.integers(1, 4))):
                    h[int(rng.integers(0, len(h)))] = a[int(rng.integers(0, len(a)))]
                if rng.random() < 0.3 and len(h) > 10:  # an indel haplotype
                    p = int(rng.integers(1, len(h) - 1))
                    h = np.concatenate([h[:p], h[p + int(rng.integers(1, 4)):]])
            haps.append(h)
        reads = []
        for _ in range(nr):
            n = int(rng.integers(*rl))
            if n <= len(root):
                s = int(rng.integers(0, len(root) - n + 1))
                b = root[s:s + n].copy()
            else:
                b = a[rng.integers(0, len(a), n)]
            flips = rng.random(n) < 0.02
            b[flips] = a[rng.integers(0, len(a), int(flips.sum()))]
            if kind == "wideq":  # the whole quality range a BAM can carry (HiFi-style Q93, gap penalties from 1)
                reads.append(Read(b, rng.integers(1, 94, n), rng.integers(1, 94, n), rng.integers(1, 94, n),
                                  rng.integers(1, 61, n)))
                continue
            qlo = 0 if kind == "lowq" else 6
            reads.append(Read(b, rng.integers(qlo, 42, n), rng.integers(6, 46, n), rng.integers(6, 46, n),
                              rng.integers(1 if kind != "lowq" else 0, 41, n)))
        regions.append((reads, haps))
    batch = RegionBatch.from_regions(regions)
    want = oracle.compute_batch(batch.as_dict(), n_threads=cores)
    # the planner's choice; the chained kernel forced at 16 lanes per pair, with 1 and with 2 or 4 streams; and at 32
    for sw in ({}, {"force_chain": 6, "force_L": 16, "force_streams": 1},
               {"force_chain": 5, "force_L": 16, "force_streams": int(rng.choice([2, 4]))},
               {"force_chain": 7, "force_L": 32}):
        with eng.switches(**sw):
            got = eng.compute(batch)
        inf = np.isinf(want)
        assert np.array_equal(np.isinf(got), inf), kind
        assert not np.isnan(got).any(), kind
        if (~inf).any():
            d = float(np.max(np.abs(got[~inf] - want[~inf])))
            worst = max(worst, d)
            assert d <= 1e-9, (kind, d)
    # f32-first mode, chained kernel forced so that the f32 sweep (and its f64 redo) really runs on these small batches
    with eng32.switches(force_chain=4, force_L=int(rng.choice([16, 16, 32]))):
        got = eng32.compute(batch)
    inf = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf), (kind, "f32 first")
    assert not np.isnan(got).any(), (kind, "f32 first")
    if (~inf).any():
        d = float(np.max(np.abs(got[~inf] - want[~inf])))
        worst32 = max(worst32, d)
        assert d <= 1e-5, (kind, "f32 first", d)
    # phmm_submit / phmm_wait: the same regions, one per submission, from four threads on one shared handle
    if batch.n_regions >= 2:
        singles = [batch.region_slice(g, g + 1) for g in range(batch.n_regions)]
        bad = []

        def worker(k):
            try:
                for g in range(k, len(singles), 4):
                    ticket, out = shared.submit(singles[g])
                    shared.wait(ticket)
                    w = want[int(batch.out_off[g]):int(batch.out_off[g + 1])]
                    fin = ~np.isinf(w)
                    assert np.array_equal(np.isinf(out), ~fin) and not np.isnan(out).any()
                    if fin.any():
                        assert float(np.max(np.abs(out[fin] - w[fin]))) <= 1e-9
            except BaseException as e:
                bad.append(e)

        th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for t in th: t.start()
        for t in th: t.join()
        assert not bad, (kind, "submit/wait", bad[0])
    n_batches += 1
    n_pairs += batch.n_out
    n_cells += batch.cells()
    kinds[kind] = kinds.get(kind, 0) + 1
print("soak ok: %d batches, %d pairs, %.3g cells, worst |hip - oracle| = %.3g (f32-first mode: %.3g), kinds %s; "
      "shared handle: %d flushes for %d submissions" % ((n_batches, n_pairs, n_cells, worst, worst32, kinds) + shared.submit_stats()))
