"""Developer tool: random regions through phmm_region_compute the way a small call goes since round 4 -- the aligner over every
(read, haplotype) pair beside pre-step and PairHMM, one kernel behind both that normalises, finds the best allele and projects
the alignment in its slot -- against the same call as the chain of round 3 (switch region_sw_all = 0), field by field; and every
read's final record against the oracle's pipeline for the best allele the device found (oracle/cigar_oracle.c, sw_oracle.c).
Region counts, PCR models, normalisation modes, soft clips and the single-allele rule are drawn per batch.
usage: python tools/soak_region.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lorikeet_amd import HipPairHMMEngine, _lib, region  # noqa: E402
from oracle import oracle  # noqa: E402
from project_scenarios import oracle_read, scenario  # noqa: E402
from test_region_hip import _cfg, _equal_calls, _noisy_quals, _priorities  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = HipPairHMMEngine(0)
t0 = time.time()
n_batches = n_reads = n_pair_calls = n_ok = 0
while time.time() - t0 < seconds:
    rng = np.random.default_rng(seed * 7919 + n_batches)
    nreg = int(rng.choice([1, 1, 1, 2, 3, 5, 9]))
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig = scenario(seed * 100003 + n_batches, n_regions=nreg, low_complexity=n_batches % 3 == 0)
    mapq = _noisy_quals(b, n_batches)
    cfg = _cfg(pcr=int(rng.integers(0, 4)), symmetric=bool(rng.integers(0, 2)), dynamic=bool(rng.integers(0, 2)))
    pri = _priorities(b, hap_cigars, ref_hap) if rng.random() < 0.7 else None
    rcfg = region.realign_config(skip_single_allele=bool(rng.integers(0, 2)))
    call = lambda: region.region_compute(eng, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig, hap_priority=pri, rcfg=rcfg)  # noqa: E731
    eng.set_switch("region_sw_all", 0)
    want = call()
    eng.set_switch("region_sw_all", 1 << 20 if n_batches % 5 else -1)  # (every fifth batch: the handle's own choice)
    before = eng.stat("region_sw_all")
    got = call()
    n_pair_calls += eng.stat("region_sw_all") - before
    _equal_calls(got, want)
    reg = np.repeat(np.arange(b.n_regions), np.diff(b.region_read_off.astype(np.int64)))
    nh = np.diff(b.region_hap_off.astype(np.int64))
    for r in range(b.n_reads):
        best = int(got.best.allele_index[r])
        if best < 0 or (rcfg.flags & _lib.PHMM_REGION_SKIP_SINGLE_ALLELE and nh[reg[r]] == 1):
            assert got.reads.status[r] == _lib.PHMM_PROJECT_UNCHANGED, (n_batches, r)
            continue
        st, pos, cig = oracle_read(b, r, reg[r], best, hap_cigars, hap_starts, ref_hap, ref_start, orig)
        assert got.reads.status[r] == st, (n_batches, r, got.reads.status[r], st)
        if st == 0:
            assert got.reads.new_pos[r] == pos and oracle.cigar_to_string(got.reads.cigars[r]) == cig, (n_batches, r)
            n_ok += 1
    n_batches += 1
    n_reads += b.n_reads
eng.set_switch("region_sw_all", -1)
print("region-call soak ok: %d batches (%d of them with every pair aligned beside the PairHMM), %d reads: equal field by field to the chain; "
      "%d realigned records equal to the oracle's for the device's best allele" % (n_batches, n_pair_calls, n_reads, n_ok))
