"""PHMM_FLAG_F32_FIRST (opt-in): large batches are swept in f32 first -- what the reference's vector arm (gkl) does --
and every read f32 cannot be trusted with is redone in f64 by the per-read kernel.  Gate for the f32 pairs: the
tolerance BASELINE.json's north_star states, 1e-5 absolute in log10 (tests/vector_pair_hmm_unit_tests.rs:63,90);
redone reads must be the f64 results bit for bit."""
import os

import numpy as np
import pytest

from lorikeet_amd import HipPairHMMEngine, synthetic
from lorikeet_amd.batch import Read, RegionBatch
from oracle import oracle

pytestmark = pytest.mark.gpu
TOL_F32 = 1e-5      # north_star / the reference's own gate for its vector path
F32_TRUST = -50.0   # log10: pairs above this are comfortably inside the f32 trust range (kernel: L*H >= 2^-196 ~ 1e-59)


@pytest.fixture(scope="module")
def engines():
    e64, e32 = HipPairHMMEngine(0), HipPairHMMEngine(0, f32_first=True)
    yield e64, e32
    e64.close()
    e32.close()


def test_f32_first_on_synthetic_regions(engines):
    e64, e32 = engines
    b = synthetic.config2(600, seed=31)
    plan = e32.plan(b)
    assert __import__("re").match(r"phmm_forward_chain_f32(_any)?<16[,>]", plan.dominant_kernel), plan.dominant_kernel
    plan.close()
    r64, r32 = e64.compute(b), e32.compute(b)
    assert np.all(r32 <= 0.0) and not np.isnan(r32).any()
    assert np.max(np.abs(r32 - r64)) <= TOL_F32
    sub = b.region_slice(0, 8)
    want = oracle.compute_batch(sub.as_dict(), n_threads=8)
    assert np.max(np.abs(r32[:int(b.out_off[8])] - want)) <= TOL_F32
    # mixed read lengths, and the stress shape (packed haplotype columns)
    for bb in (synthetic.config3(400, seed=32), synthetic.config5(8, seed=33)):
        assert np.max(np.abs(e32.compute(bb) - e64.compute(bb))) <= TOL_F32
    # haplotypes of 600 columns: 32 lanes per pair
    long_haps = synthetic.make_regions(400, 64, 8, 600, 150, seed=36)
    plan = e32.plan(long_haps)
    assert __import__("re").match(r"phmm_forward_chain_f32(_any)?<32[,>]", plan.dominant_kernel), plan.dominant_kernel
    plan.close()
    r32 = e32.compute(long_haps)
    assert np.max(np.abs(r32 - e64.compute(long_haps))) <= TOL_F32
    want = oracle.compute_batch(long_haps.region_slice(0, 4).as_dict(), n_threads=8)
    assert np.max(np.abs(r32[:int(long_haps.out_off[4])] - want)) <= TOL_F32


def test_f32_first_known_answer_vectors(engines, kat_rows):
    """The reference's 104 vectors, one region per haplotype so that the reads chain (forced: the batch is tiny)."""
    _, _ = engines
    by_hap = {}
    for r in kat_rows:
        by_hap.setdefault(r["hap"], []).append(r)
    regs = [([Read(r["read"], r["qual"], r["ins"], r["dele"], r["gcp"]) for r in rows], [hap]) for hap, rows in by_hap.items()]
    kb = RegionBatch.from_regions(regs)
    exp = np.array([r["expected"] for rows in by_hap.values() for r in rows])
    for lanes, streams in (("16", "1"), ("16", "2"), ("16", "4"), ("32", "1")):
        eng = HipPairHMMEngine(0, f32_first=True)
        with eng.switches(force_chain=5, force_L=int(lanes), force_streams=int(streams)):
            plan = eng.plan(kb)
            assert __import__("re").match(r"phmm_forward_chain_f32(_any)?<%s[,>]" % lanes, plan.dominant_kernel), plan.dominant_kernel
            plan.close()
            got = eng.compute(kb)
        eng.close()
        assert np.max(np.abs(got - exp)) < TOL_F32


def test_what_f32_cannot_be_trusted_with_is_redone_in_f64(engines):
    e64, e32 = engines
    rng = np.random.default_rng(34)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    regs = []
    for g in range(1500):
        root = alpha[rng.integers(0, 4, 200)]
        haps = [root]
        for _ in range(3):                  # the other haplotypes: the root with two SNVs
            hh = root.copy()
            hh[rng.integers(0, 200, 2)] = alpha[rng.integers(0, 4, 2)]
            haps.append(hh)
        if g % 50 == 0:
            haps[1] = haps[1].copy()
            haps[1][17] = ord("N")          # general path: the whole run goes to the f64 kernel
        reads = []
        for i in range(40):
            if i % 3 == 0:
                bases = alpha[rng.integers(0, 4, 100)]   # unrelated to every haplotype: log10 L ~ -150, below f32's range
            else:
                s = int(rng.integers(0, 100))
                bases = haps[i % 4][s:s + 100].copy()
            gcp = np.full(100, 10, np.uint8)
            if g % 77 == 0 and i == 5:
                gcp[3] = 0                  # cannot be pre-scaled: general path
            reads.append(Read(bases, np.full(100, 30, np.uint8), np.full(100, 40, np.uint8), np.full(100, 40, np.uint8), gcp))
        regs.append((reads, haps))
    b = RegionBatch.from_regions(regs)
    plan = e32.plan(b)
    assert __import__("re").match(r"phmm_forward_chain_f32(_any)?<16[,>]", plan.dominant_kernel), plan.dominant_kernel
    plan.close()
    r64, r32 = e64.compute(b), e32.compute(b)
    with e64.switches(force_chain=0):                   # the f64 per-read kernel: what the redo pass runs
        per_read = e64.compute(b).reshape(-1, 4)
    m = r64.reshape(-1, 4)
    m32 = r32.reshape(-1, 4)
    redone = (m.min(axis=1) < F32_TRUST - 25)          # reads with a pair far below the trust range are redone entirely
    assert redone.sum() > 15000
    assert np.array_equal(m32[redone], per_read[redone])  # ... and are that kernel's f64 results, bit for bit
    assert np.max(np.abs(r32 - r64)) <= TOL_F32
    ok = m.min(axis=1) > F32_TRUST                     # reads the f32 sweep keeps: not identical, within the gate
    assert ok.sum() > 15000 and 0 < np.max(np.abs(m32[ok] - m[ok])) <= TOL_F32
    want = oracle.compute_batch(b.region_slice(0, 40).as_dict(), n_threads=8)
    assert np.max(np.abs(r32[:int(b.out_off[40])] - want)) <= TOL_F32


def test_small_batches_stay_in_f64(engines):
    e64, e32 = engines
    b = synthetic.config2(6, seed=35)   # not enough work to chain: per-read f64 kernel in both modes
    assert np.array_equal(e32.compute(b), e64.compute(b))
