"""Pins the engine-level oracle (oracle/engine_oracle.c) to the known answers SURVEY.md section 4 derives
from the reference's own engine fixture (tests/pair_hmm_likelihood_calculation_engine_unit_tests.rs:21-88:
one 10 x 'A' read, Q30, MAPQ 60, against 11 x 'A' (ref) and the same with a C at index 5; gcp 93,
PCRErrorModel::Conservative, threshold 16) and to hand-checkable properties of each step."""
import math

import numpy as np

from lorikeet_amd.batch import Read, RegionBatch
from oracle import oracle


def test_pcr_error_model_cache():
    c = oracle.pcr_error_model_cache("conservative")  # engine.rs:169-193
    assert list(c[:13]) == [40, 39, 39, 39, 39, 39, 39, 38, 38, 38, 38, 37, 37]
    assert c[100] == 6 and all(c[i] >= c[i + 1] for i in range(100)) and min(c) == 6
    assert not oracle.pcr_error_model_cache("none").any()
    h, a = oracle.pcr_error_model_cache("hostile"), oracle.pcr_error_model_cache("aggressive")
    assert all(h[i] <= a[i] <= c[i] for i in range(101))  # more aggressive -> lower quals


def test_tandem_repeat_lengths():
    lib = oracle.lib()

    def rl(s, off):
        a, p = oracle._u8(s)
        return lib.oracle_find_tandem_repeat_length(p, len(a), off)
    assert [rl(b"A" * 10, i) for i in range(9)] == [10] * 9        # inside a homopolymer: bw + fw
    assert rl(b"ACGTACGTAC", 3) == 1                                  # no unit repeats twice on either side of offset 3
    assert rl(b"GATTACA", 0) == 1 and rl(b"GATTACA", 2) == 2         # TT
    assert rl(b"TTCTTCCCC", 5) == 4                                   # engine.rs:589-592: (C)4, not (TTC)2
    assert rl(b"A" * 150, 75) == 100                                  # capped at MAX_REPEAT_LENGTH


def test_engine_fixture_known_answers():
    q, i, d = oracle.modify_read_qualities("conservative", b"A" * 10, 60, [30] * 10, [45] * 10, [45] * 10, 16)
    assert list(q) == [30] * 10
    assert list(i) == [38] * 9 + [45] and list(d) == [38] * 9 + [45]  # the loop never touches the last base
    rd = Read(b"A" * 10, q, i, d, [93] * 10)
    b = RegionBatch.from_regions([([rd], [b"A" * 11, b"AAAAACAAAAA"])])
    raw = oracle.compute_batch(b.as_dict())
    assert abs(raw[0] - -0.7446794209931795) < 1e-12 and abs(raw[1] - -4.128855297736436) < 1e-12
    cap = oracle.lib().oracle_qual_to_prob  # noqa: F841 (keep lib loaded)
    cap = (45 * -0.1) * math.log10(math.e)   # log_to_log10(qual_to_error_prob_log10(45))
    assert abs(cap - -1.9543251685646332) < 1e-15
    v = oracle.normalize_likelihoods(raw.reshape(1, 2).T.copy(), cap, True, 0)
    assert abs(v[0, 0] - -0.7446794209931795) < 1e-12 and abs(v[1, 0] - -2.6990045895578127) < 1e-12
    thr = oracle.read_disqualification_threshold([30] * 10, False, 1.0, 0.02)
    assert thr == -4.0                                                 # min(2, ceil(10 * 0.02)) * -4
    v2, keep, n = oracle.filter_poorly_modeled_evidence(v, [thr])
    assert n == 1 and keep.tolist() == [True] and v2[0, 0] > v2[1, 0]  # L[ref] > L[alt]: the reference's assertion


def test_cap_minimum_read_qualities():
    q, i, d = oracle.modify_read_qualities("none", b"ACGTACGT", 20, [40, 17, 18, 5, 30, 19, 20, 21], [3, 6, 45, 0, 7, 5, 6, 99],
                                           [45, 2, 6, 6, 5, 45, 1, 7], 18)
    assert list(q) == [20, 6, 18, 6, 20, 19, 20, 20]   # min(q, mapq) then < 18 -> 6
    assert list(i) == [6, 6, 45, 6, 7, 6, 6, 99] and list(d) == [45, 6, 6, 6, 6, 45, 6, 7]
    q, _, _ = oracle.modify_read_qualities("none", b"ACGT", 20, [40, 17, 18, 5], [45] * 4, [45] * 4, 18, True)
    assert list(q) == [40, 6, 18, 6]                   # --disable-cap-base-qualities-to-map-quality


def test_thresholds():
    assert oracle.read_disqualification_threshold([30] * 150, False, 1.0, 0.02) == -8.0   # min(2, ceil(3)) * -4
    dyn = oracle.read_disqualification_threshold([30] * 150, True, 1.0, 0.02)
    want = -0.1 * (150 * 0.039111985 + math.sqrt(150 * 1.207526336))
    assert abs(dyn - min(want, -12.0)) < 1e-12          # min(dynamic, ceil(150*0.02) * -4)
    lo = oracle.read_disqualification_threshold([0, 1, 2, 45, 200], True, 2.0, 0.001)  # quals clamp to table rows 1..40
    t = [(5.996842844, 0.196616587)] * 2 + [(5.870018422, 1.388545569)] + [(0.004911394, 0.200422214)] * 2
    want = -0.1 * (sum(m for m, _ in t) + 2.0 * math.sqrt(sum(v for _, v in t)))
    assert abs(lo - min(want, -4.0)) < 1e-12


def test_normalize_and_filter_semantics():
    nan = float("nan")
    m = np.array([[-1.0, -9.0, -3.0], [-5.0, -2.0, -3.5], [-20.0, -2.5, -30.0]])  # [allele, read]
    v = oracle.normalize_likelihoods(m.copy(), -3.0, True, 0)
    assert np.array_equal(v, [[-1.0, -5.0, -3.0], [-4.0, -2.0, -3.5], [-4.0, -2.5, -6.0]])
    # asymmetric: best is searched among non-reference alleles only (allele_likelihoods.rs:479-494)
    v = oracle.normalize_likelihoods(m.copy(), -3.0, False, 0)
    assert np.array_equal(v, [[-1.0, -5.0, -3.0], [-5.0, -2.0, -3.5], [-8.0, -2.5, -6.5]])
    v = oracle.normalize_likelihoods(m.copy(), -math.inf, True, 0)
    assert np.array_equal(v, m)                                                      # infinite cap: no-op
    one = oracle.normalize_likelihoods(m[:1].copy(), -3.0, True, 0)
    assert np.array_equal(one, m[:1])                                                # single allele: no-op
    v, keep, n = oracle.filter_poorly_modeled_evidence(m.copy(), [-0.5, -2.0, -2.9])
    assert keep.tolist() == [False, True, False] and n == 1
    assert np.array_equal(v[:, 0], m[:, 1]) and np.isnan(v[:, 1:]).all()
    del nan


# ---- the reference's own property tests of the container, restated against the engine oracle ----------------------
def _shapes():
    """(alleles, reads per sample) like data_for_test_*: a few sample / allele / read-count combinations."""
    rng = np.random.default_rng(13)
    return [(int(rng.integers(1, 7)), [int(rng.integers(0, 40)) for _ in range(int(rng.integers(1, 4)))]) for _ in range(25)]


def test_filter_poorly_modeled_reads_property():
    """tests/allele_likelihoods_unit_tests.rs:399-442 (+ make_good_and_bad_likelihoods :560-581): every odd read gets
    -10000 for all alleles, threshold -100 for everybody -> exactly the even reads survive, in order, with their values;
    the removed ones are all accounted for."""
    rng = np.random.default_rng(7)
    for n_alleles, per_sample in _shapes():
        for n_reads in per_sample:
            original = rng.uniform(-50.0, 0.0, (n_alleles, n_reads))   # "good" reads: anything above the threshold
            original[:, 1::2] = -10000.0
            v, keep, n_kept = oracle.filter_poorly_modeled_evidence(original.copy(), [-100.0] * n_reads)
            assert n_kept == (n_reads + 1) // 2
            assert keep.tolist() == [(r & 1) == 0 for r in range(n_reads)]
            for r in range(n_kept):
                assert np.array_equal(v[:, r], original[:, 2 * r])      # compacted, in order, values untouched
            assert np.isnan(v[:, n_kept:]).all()                        # the slots of the removed reads


def test_normalize_cap_worst_likelihood_property():
    """tests/allele_likelihoods_unit_tests.rs:725-770: normalize_likelihoods(-0.001, symmetric) == max(best - 0.001, v)
    per read, and a read whose best is -inf is left alone."""
    rng = np.random.default_rng(8)
    for n_alleles, per_sample in _shapes():
        for n_reads in per_sample:
            if n_reads == 0:
                continue
            original = -np.abs(rng.normal(0.0, 3.0, (n_alleles, n_reads)))
            if n_reads > 2:
                original[:, 1] = -np.inf
            got = oracle.normalize_likelihoods(original.copy(), -0.001, True, 0)
            want = original.copy()
            if n_alleles > 1:   # a single allele is a no-op in the reference (allele_likelihoods.rs:386-389)
                for r in range(n_reads):
                    best = original[:, r].max()
                    if best != -np.inf:
                        want[:, r] = np.maximum(best - 0.001, original[:, r])
            assert np.array_equal(got, want)
