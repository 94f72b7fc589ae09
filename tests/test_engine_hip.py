"""Engine-level parity on the GPU: phmm_engine_compute (pre-step, PairHMM, normalise, disqualification
decision) against the oracle pipeline, and the reference's own engine test
(tests/pair_hmm_likelihood_calculation_engine_unit_tests.rs:21-88) through the mirrored interface."""
import math

import numpy as np
import pytest

from lorikeet_amd.batch import Read, RegionBatch
from lorikeet_amd.likelihood_engine import (AssemblyResultSet, AVXMode, PairHMMLikelihoodCalculationEngine,
                                            PCRErrorModel, log_to_log10, qual_to_error_prob_log10)
from lorikeet_amd.pair_hmm import Haplotype, HmmRead
from oracle import oracle

pytestmark = pytest.mark.gpu
MODELS = {PCRErrorModel.NONE: "none", PCRErrorModel.HOSTILE: "hostile", PCRErrorModel.AGGRESSIVE: "aggressive",
          PCRErrorModel.CONSERVATIVE: "conservative"}


def test_compute_likelihoods_reference_engine_test():
    """The reference's test, argument for argument."""
    lce = PairHMMLikelihoodCalculationEngine(
        93, log_to_log10(qual_to_error_prob_log10(45)), PCRErrorModel.CONSERVATIVE, 16, False, 1.0, 0.02, True, False,
        True, AVXMode.Hip)
    n = 10
    read1 = HmmRead(b"A" * n, [30] * n, mapq=60)   # create_artificial_read_default("test", 0, 0, 10): 10 x 'A', Q30
    per_sample_read_list = {0: [read1]}
    ref_bases = b"A" * (n + 1)
    hap1 = Haplotype(ref_bases, True)
    assembly_result_set = AssemblyResultSet(hap1)
    assembly_result_set.add_haplotype(hap1)
    hap2 = Haplotype(ref_bases[:5] + b"C" + ref_bases[6:], False)
    assembly_result_set.add_haplotype(hap2)
    likes = lce.compute_read_likelihoods(assembly_result_set, [0], per_sample_read_list)
    assert len(likes.alleles) == 2
    assert likes.evidence_count() == 1
    v1, v2 = likes.sample_matrix(0)[0, 0], likes.sample_matrix(0)[1, 0]
    assert v1 > v2, "Matching hap should have a higher likelihood"
    # SURVEY.md section 4 known answers for this fixture
    assert abs(v1 - -0.7446794209931795) < 1e-9 and abs(v2 - -2.6990045895578127) < 1e-9


def _oracle_pipeline(cfg, regions):
    """engine.rs:195-242 with the oracle's pieces; returns per region (normalised [read][hap], keep)."""
    res = []
    for reads, haps in regions:
        mod, thr = [], []
        for r in reads:
            q, i, d = oracle.modify_read_qualities(MODELS[cfg["pcr"]], r.bases, r.mapq, r.quals,
                                                   r.base_insertion_qualities(), r.base_deletion_qualities(),
                                                   cfg["bq_threshold"], cfg["disable_cap"])
            mod.append(Read(r.bases, q, i, d, np.full(len(r), cfg["gcp"], np.uint8)))
            thr.append(oracle.read_disqualification_threshold(r.quals, cfg["dynamic"], cfg["scale"], cfg["err"]))
        b = RegionBatch.from_regions([(mod, [h.get_bases() for h in haps])])
        raw = oracle.compute_batch(b.as_dict(), n_threads=4).reshape(len(reads), len(haps))
        ref = next((j for j, h in enumerate(haps) if h.is_ref), None)
        norm = oracle.normalize_likelihoods(raw.T.copy(), cfg["cap"], cfg["symmetric"], ref)
        _, keep, _ = oracle.filter_poorly_modeled_evidence(norm.copy(), thr)
        res.append((norm.T, keep))
    return res


def _random_regions(rng, n_regions, with_tags):
    alpha = np.frombuffer(b"ACGT", np.uint8)
    regions = []
    for _ in range(n_regions):
        nh = int(rng.integers(1, 7))
        root = alpha[rng.integers(0, 4, int(rng.integers(60, 260)))]
        # low-complexity stretches so the PCR model has tandem repeats to find
        for _ in range(3):
            s = int(rng.integers(0, len(root) - 30))
            unit = alpha[rng.integers(0, 4, int(rng.integers(1, 5)))]
            rep = np.tile(unit, 12)[:int(rng.integers(6, 25))]
            root[s:s + len(rep)] = rep[:len(root) - s]
        haps = []
        for j in range(nh):
            h = root.copy()
            for _ in range(int(rng.integers(0, 3)) if j else 0):
                h[int(rng.integers(0, len(h)))] = alpha[int(rng.integers(0, 4))]
            hh = Haplotype(bytes(h), is_ref=(j == int(rng.integers(0, nh))))
            if hh not in haps:
                haps.append(hh)
        reads = []
        for _ in range(int(rng.integers(0, 12))):
            n = int(rng.integers(1, min(120, len(root))))
            s = int(rng.integers(0, len(root) - n + 1))
            bases = root[s:s + n].copy()
            flips = rng.random(n) < 0.03
            bases[flips] = alpha[rng.integers(0, 4, int(flips.sum()))]
            quals = rng.choice([2, 6, 12, 17, 18, 22, 27, 32, 37, 41], n)
            ins = rng.integers(0, 60, n) if with_tags else None
            dele = rng.integers(0, 60, n) if with_tags else None
            reads.append(HmmRead(bytes(bases), quals, ins, dele, mapq=int(rng.choice([0, 10, 29, 60]))))
        regions.append((reads, haps))
    return regions


@pytest.mark.parametrize("pcr", list(MODELS))
@pytest.mark.parametrize("dynamic,symmetric,with_tags", [(False, True, False), (True, False, True), (True, True, True)])
def test_engine_matches_oracle_pipeline(pcr, dynamic, symmetric, with_tags):
    cfg = dict(gcp=10, cap=-4.5 * math.log10(math.e), pcr=pcr, bq_threshold=18, dynamic=dynamic, scale=1.0, err=0.02,
               symmetric=symmetric, disable_cap=(pcr == PCRErrorModel.HOSTILE))
    eng = PairHMMLikelihoodCalculationEngine(cfg["gcp"], cfg["cap"], pcr, cfg["bq_threshold"], dynamic, cfg["scale"],
                                             cfg["err"], symmetric, cfg["disable_cap"])
    regions = _random_regions(np.random.default_rng(100 + int(pcr) + 10 * dynamic), 12, with_tags)
    got = eng.compute_regions(regions)
    want = _oracle_pipeline(cfg, regions)
    n_removed = 0
    for (gm, gk), (wm, wk) in zip(got, want):
        assert gm.shape == wm.shape
        if gm.size:
            assert np.max(np.abs(gm - wm)) <= 1e-9
        assert np.array_equal(gk, wk)
        n_removed += int((~wk).sum())
    if dynamic:
        assert n_removed > 0  # the test data does exercise the removal branch


def test_engine_surface_filters_and_compacts():
    """compute_read_likelihoods moves poorly modelled reads to filtered evidence, compacts the [allele, read]
    matrix and pads with NaN (allele_likelihoods.rs:968-1018)."""
    eng = PairHMMLikelihoodCalculationEngine(10, -4.5 * math.log10(math.e), PCRErrorModel.CONSERVATIVE, 18, True, 1.0,
                                             0.02, True, False)
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    ref = alpha[rng.integers(0, 4, 120)]
    alt = ref.copy(); alt[60] = alpha[(np.where(alpha == alt[60])[0][0] + 1) % 4]
    ars = AssemblyResultSet(Haplotype(bytes(ref), True))
    ars.add_haplotype(Haplotype(bytes(alt), False))
    good = [HmmRead(bytes(ref[s:s + 50]), [30] * 50, mapq=60) for s in (5, 30, 60)]
    junk = [HmmRead(bytes(alpha[rng.integers(0, 4, 50)]), [30] * 50, mapq=60) for _ in range(2)]
    likes = eng.compute_read_likelihoods(ars, [0, 1], {0: [good[0], junk[0], good[1]], 1: [junk[1], good[2]]})
    assert [len(likes.evidence_by_sample_index[s]) for s in (0, 1)] == [2, 1]
    assert likes.filtered_evidence_by_sample_index[0] == [junk[0]] and likes.filtered_evidence_by_sample_index[1] == [junk[1]]
    m0, m1 = likes.sample_matrix(0), likes.sample_matrix(1)
    assert m0.shape == (2, 3) and m1.shape == (2, 2)
    assert not np.isnan(m0[:, :2]).any() and np.isnan(m0[:, 2]).all() and np.isnan(m1[:, 1]).all()
    assert (m0[0, :2] >= m0[1, :2]).all()  # reads copied from the reference prefer it


def test_large_engine_call_is_pipelined_transparently():
    """A call big enough to be cut into chunks (three (arena, stream) slots, staging of chunk i+1 overlapping the
    kernels of chunk i) returns exactly what the one-shot path returns: regions are independent."""
    import ctypes as C
    import os

    from lorikeet_amd import HipPairHMMEngine, _lib, synthetic
    b = synthetic.config2(600, seed=77)  # 11.5 MB per per-base array: three chunks
    eng = HipPairHMMEngine(0)
    cfg = _lib.EngineConfig()
    cfg.constant_gcp, cfg.pcr_error_model, cfg.base_quality_score_threshold = 10, 3, 18
    cfg.dynamic_read_disqualification, cfg.symmetrically_normalize_alleles_to_reference = 1, 1
    cfg.log10_global_read_mismapping_rate = -4.5 * math.log10(math.e)
    cfg.read_disqualification_scale, cfg.expected_error_rate_per_base = 1.0, 0.02
    mapq = np.full(b.n_reads, 60, np.uint8)
    mapq[::7] = 20
    ref = np.zeros(b.n_regions, np.int32)
    pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731

    def run():
        out = np.empty(b.n_out, np.float64)
        keep = np.zeros(b.n_reads, np.uint8)
        st = eng.lib.phmm_engine_compute(
            eng._h, C.byref(cfg), b.n_regions, pp(b.region_read_off, _lib.u32p), pp(b.region_hap_off, _lib.u32p),
            pp(b.read_off, _lib.u32p), pp(b.read_bases, _lib.u8p), pp(b.base_q, _lib.u8p), None, None, pp(mapq, _lib.u8p),
            pp(b.hap_off, _lib.u32p), pp(b.hap_bases, _lib.u8p), pp(ref, C.POINTER(C.c_int32)), pp(b.out_off, _lib.u64p),
            pp(out, _lib.f64p), pp(keep, _lib.u8p))
        assert st == 0, eng.last_error()
        return out, keep

    out_p, keep_p = run()
    with eng.switches(no_pipeline=1):
        out_1, keep_1 = run()
    # chunks plan their own shapes (run lengths differ with the batch size): same numbers up to the last-row
    # summation order, identical keep decisions
    assert np.max(np.abs(out_p - out_1)) <= 1e-12
    assert np.array_equal(keep_p, keep_1)
    assert 0.9 < keep_p.mean() <= 1.0
    eng.close()


def _engine_call(eng, batch, mapq, cap, symmetric=True, ref=None, dynamic=False, pcr=0, gcp=10, bq_threshold=18):
    import ctypes as C
    from lorikeet_amd import _lib
    cfg = _lib.EngineConfig()
    cfg.constant_gcp, cfg.pcr_error_model, cfg.base_quality_score_threshold = gcp, pcr, bq_threshold
    cfg.dynamic_read_disqualification, cfg.symmetrically_normalize_alleles_to_reference = int(dynamic), int(symmetric)
    cfg.log10_global_read_mismapping_rate = cap
    cfg.read_disqualification_scale, cfg.expected_error_rate_per_base = 1.0, 0.02
    out = np.empty(batch.n_out, np.float64)
    keep = np.zeros(batch.n_reads, np.uint8)
    mq = np.full(batch.n_reads, mapq, np.uint8)
    pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
    st = eng.lib.phmm_engine_compute(
        eng._h, C.byref(cfg), batch.n_regions, pp(batch.region_read_off, _lib.u32p), pp(batch.region_hap_off, _lib.u32p),
        pp(batch.read_off, _lib.u32p), pp(batch.read_bases, _lib.u8p), pp(batch.base_q, _lib.u8p), pp(batch.ins_q, _lib.u8p),
        pp(batch.del_q, _lib.u8p), pp(mq, _lib.u8p), pp(batch.hap_off, _lib.u32p), pp(batch.hap_bases, _lib.u8p),
        pp(ref, C.POINTER(C.c_int32)) if ref is not None else None, pp(batch.out_off, _lib.u64p), pp(out, _lib.f64p), pp(keep, _lib.u8p))
    assert st == 0, eng.last_error()
    return out, keep


def test_filter_and_normalise_properties_of_the_reference_on_the_device(hip_engine):
    """The reference's container properties (tests/allele_likelihoods_unit_tests.rs:399-442, :725-770) through the HIP
    engine call: with every odd read unrelated to the haplotypes, exactly the even reads are kept, in order; and with
    a cap of -0.001 every value is max(best - 0.001, raw) of the PairHMM results."""
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    regions = []
    for g in range(6):
        root = acgt[rng.integers(0, 4, 200)]
        haps = [root.copy() for _ in range(4)]
        for h in haps[1:]:
            h[rng.integers(0, 200, 2)] = acgt[rng.integers(0, 4, 2)]
        reads = []
        for r in range(24):
            n = int(rng.choice([80, 120]))
            if r & 1:                                      # odd reads: random sequence, likelihood ~ 1e-100 and worse
                bases = acgt[rng.integers(0, 4, n)]
            else:                                          # even reads: a piece of a haplotype with at most one error
                s0 = int(rng.integers(0, 200 - n))
                bases = haps[int(rng.integers(0, 4))][s0:s0 + n].copy()
                if r % 4 == 0:
                    bases[n // 2] = acgt[(int(np.where(acgt == bases[n // 2])[0][0]) + 1) % 4]
            # PCR model off, quals above the cap threshold, MAPQ 60: the pre-step leaves the qualities alone
            reads.append(Read(bases, np.full(n, 35), np.full(n, 40), np.full(n, 40), np.full(n, 10)))
        regions.append((reads, haps))
    bad = RegionBatch.from_regions(regions)
    raw = hip_engine.compute(bad)                          # plain PairHMM on the same (unmodified) qualities
    out, keep = _engine_call(hip_engine, bad, 60, -0.001)
    assert keep.tolist() == [1 - (r & 1) for r in range(bad.n_reads)]     # threshold min(2, ceil(R * 0.02)) * -4 = -8
    for g in range(bad.n_regions):
        r0, r1 = int(bad.region_read_off[g]), int(bad.region_read_off[g + 1])
        nh = int(bad.region_hap_off[g + 1] - bad.region_hap_off[g])
        m_raw = raw[int(bad.out_off[g]):int(bad.out_off[g + 1])].reshape(r1 - r0, nh)
        m_out = out[int(bad.out_off[g]):int(bad.out_off[g + 1])].reshape(r1 - r0, nh)
        want = np.maximum(m_raw.max(axis=1, keepdims=True) - 0.001, m_raw)
        assert np.max(np.abs(m_out - want)) <= 1e-12
        # scatter like the caller does: kept reads in order == the even reads, values equal
        kept = m_out[keep[r0:r1] == 1]
        assert np.array_equal(kept, m_out[0::2])
    # the oracle pipeline agrees on both (normalised values and keep flags)
    for g in range(2):
        sub = bad.region_slice(g, g + 1)
        raw_o = oracle.compute_batch(sub.as_dict(), n_threads=4).reshape(sub.n_reads, -1)
        norm = oracle.normalize_likelihoods(raw_o.T.copy(), -0.001, True, 0)
        thr = [oracle.read_disqualification_threshold(sub.base_q[int(sub.read_off[r]):int(sub.read_off[r + 1])], False, 1.0, 0.02)
               for r in range(sub.n_reads)]
        _, k_o, _ = oracle.filter_poorly_modeled_evidence(norm.copy(), thr)
        r0 = int(bad.region_read_off[g])
        assert k_o.tolist() == [bool(x) for x in keep[r0:r0 + sub.n_reads]]
        assert np.max(np.abs(norm.T.reshape(-1) - out[int(bad.out_off[g]):int(bad.out_off[g + 1])])) <= 1e-9


def test_ragged_mix_matches_the_oracle(hip_engine):
    """regions drawn from a long-tailed distribution (3 ... 5 000 reads, 1 ... 128 haplotypes, H 60 ... 500,
    R 30 ... 250 mixed, some haplotypes with 'N'): 70 regions of the bench's ragged set against the oracle, resident and
    through host buffers, plus the whole 1 536-region set resident vs host buffers (one launch per lanes-per-pair value vs
    chunks with plans of their own)."""
    from lorikeet_amd import sharding, synthetic
    full = synthetic.ragged()
    cells = sharding.region_cells(full)
    assert cells.max() > 1e9 and cells.min() < 1e5 and (full.hap_bases == ord("N")).sum() > 50
    small = [g for g in range(full.n_regions) if cells[g] < 6e7][:68] + [int(np.argsort(cells)[-40])]   # + one heavy region
    sub = sharding.take_regions(full, small)
    want = oracle.compute_batch(sub.as_dict(), n_threads=16)
    got = hip_engine.compute(sub)
    assert np.max(np.abs(got - want)) <= 1e-9
    plan = hip_engine.plan(sub)
    plan.upload()
    plan.launch()
    assert np.max(np.abs(plan.download() - want)) <= 1e-9
    plan.close()
    plan = hip_engine.plan(full)
    # one per lanes-per-pair value and range of K -- four ranges, side by side on parallel streams -- (+ per-read classes),
    # not one per <K, streams> class (72 in round 1)
    assert plan.num_launches <= 10, plan.num_launches
    plan.upload()
    plan.launch()
    resident = plan.download()
    plan.close()
    host = hip_engine.compute(full)
    assert np.max(np.abs(resident - host)) <= 1e-9 and (resident <= 0).all()


def test_one_region_per_call_equals_the_region_inside_a_large_batch():
    """A call of one region does without the copy engine (inputs fetched from the pinned mirror by a kernel, keep flags,
    normalised likelihoods and the status word stored into it by the post-step); a large batch goes through copies and
    the chunked host path.  Same keep flags, same likelihoods (other kernel shapes: 1e-12)."""
    eng = PairHMMLikelihoodCalculationEngine(10, -4.5 * math.log10(math.e), PCRErrorModel.CONSERVATIVE, 18, True, 1.0, 0.02, True, False)
    regions = _random_regions(np.random.default_rng(31), 40, True)
    filler = _random_regions(np.random.default_rng(32), 6, False)
    big = eng.compute_regions(regions + filler * 400)
    removed = 0
    for k, reg in enumerate(regions):
        (m, keep), = eng.compute_regions([reg])
        assert m.shape == big[k][0].shape and np.array_equal(keep, big[k][1]), k
        if m.size:
            assert np.max(np.abs(m - big[k][0])) <= 1e-12, k
        removed += int((~keep).sum())
    assert removed > 0
