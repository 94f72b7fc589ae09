"""oracle_best_alleles (oracle/engine_oracle.c) restates AlleleLikelihoods::search_best_allele + BestAllele::new
(src/model/allele_likelihoods.rs:457-554, :1142-1160).  Pinned by the reference's own property test
(tests/allele_likelihoods_unit_tests.rs:250-365, data_for_test_best_alleles :802-813): random N(0,1) likelihoods, priority
1 for the reference allele and 0 otherwise -- the best allele is the arg-max unless the reference is within 0.2 of it,
in which case the reference takes over, with likelihood and confidence to match."""
import numpy as np
import pytest

from oracle import oracle

THR = 0.2  # LOG_10_INFORMATIVE_THRESHOLD, allele_likelihoods.rs:17


@pytest.mark.parametrize("n_alleles,ref_index", [(1, 0), (2, 0), (2, 1), (3, None), (5, 2), (8, 0), (17, 16)])
def test_reference_property_ref_override(n_alleles, ref_index):
    rng = np.random.default_rng(n_alleles * 31 + (ref_index or 0))
    v = rng.normal(0.0, 1.0, size=(n_alleles, 400))
    pri = np.array([1 if a == ref_index else 0 for a in range(n_alleles)], np.int32)
    best, lk, conf = oracle.best_alleles(v, pri, THR)
    for r in range(v.shape[1]):
        # the test's own scan (:285-305)
        b, blk, slk = None, -np.inf, -np.inf
        for a in range(n_alleles):
            if v[a, r] > blk:
                slk, blk, b = blk, v[a, r], a
            elif v[a, r] > slk:
                slk = v[a, r]
        ref_lk = v[ref_index, r] if ref_index is not None else -np.inf
        override = ref_index is not None and ref_index != b and blk - ref_lk < THR
        assert best[r] == (ref_index if override else b)                                   # :340-350
        assert lk[r] == pytest.approx(ref_lk if override else blk, abs=1e-12)              # :326-338
        assert conf[r] == pytest.approx(ref_lk - blk if override else blk - slk, abs=1e-12)  # :352-360


def test_no_priorities_is_the_plain_argmax_and_ties_keep_the_first():
    v = np.array([[-1.0, -2.0, -3.0, -1.0], [-1.0, -1.5, -3.0, -0.5], [-4.0, -1.5, -3.0, -0.5]])
    best, lk, conf = oracle.best_alleles(v, None, THR)
    assert best.tolist() == [0, 1, 0, 1] and lk.tolist() == [-1.0, -1.5, -3.0, -0.5] and conf.tolist() == [0.0, 0.0, 0.0, 0.0]


def test_priorities_choose_among_everything_within_the_threshold():
    # alleles 0, 2, 3 lie within 0.2 of the best (allele 3); priorities pick allele 2; allele 1 is too far to matter
    v = np.array([[-1.10], [-2.00], [-1.15], [-1.00]])
    best, lk, conf = oracle.best_alleles(v, [0, 9, 5, -1], THR)
    assert best[0] == 2 and lk[0] == -1.15
    # the runner-up is the allele the second-highest priority names (allele 0), confidence = difference of their likelihoods
    assert conf[0] == pytest.approx(-1.15 - (-1.10))


def test_no_alleles_gives_none():
    best, lk, conf = oracle.best_alleles(np.zeros((0, 3)), None, THR)
    assert best.tolist() == [-1, -1, -1] and np.all(np.isneginf(lk)) and np.all(np.isnan(conf))
