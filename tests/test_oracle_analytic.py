"""The reference's analytic PairHMM tests (tests/pair_hmm_unit_tests.rs, tests/pair_hmm_model_unit_tests.rs)
restated against the CPU oracle.  These pin the oracle beyond the 104 KATs: SNP/indel expectations,
mismatch at every position, all-matching reads, big reads, caching == full recompute."""
import math

import numpy as np
import pytest

import reference_cases as rc
from oracle import oracle


def _run(cases, threads=4):
    cases = list(cases)
    got = oracle.compute_batch(rc.to_batch(cases).as_dict(), disable_tristate=True, n_threads=threads)
    assert len(got) == len(cases)
    for c, g in zip(cases, got):
        rc.check(c, float(g))
    return got


def test_basic_likelihoods():
    got = _run(rc.basic_likelihood_cases(extensive=True))
    assert len(got) == 5 * 4 * 3 * (16 + 11 * 4 * 2 * 4)


def test_mismatch_in_every_position():
    _run(rc.mismatch_every_position_cases())


def test_hmm_providers_and_big_reads():
    _run(rc.hmm_provider_cases())
    _run(rc.big_read_cases())
    for _ in range(3):
        _run([rc.max_lengths_case()])


def test_find_first_position_where_haplotypes_differ():
    # pair_hmm_unit_tests.rs:685-723
    lib = oracle.lib()
    for n1 in range(10, 30, 3):
        for n2 in range(10, 50, 7):
            for site in range(0, max(n1, n2) + 1, 2):
                for one_is_diff in (True, False):
                    h1, h2 = bytearray(b"A" * n1), bytearray(b"A" * n2)
                    tgt = h1 if one_is_diff else h2
                    expected = min(n1, n2)
                    if site < len(tgt):
                        tgt[site] = ord("C")
                        expected = min(site, min(n1, n2))
                    a1, p1 = oracle._u8(bytes(h1))
                    a2, p2 = oracle._u8(bytes(h2))
                    assert lib.oracle_find_first_position_where_haplotypes_differ(p1, n1, p2, n2) == expected


def test_haplotype_prefix_caching_equals_full_recompute():
    # pair_hmm_unit_tests.rs:725-814, tolerance 1e-9
    prefix, (root_1, root_2, root_3), reads = rc.haplotype_indexing_inputs()
    for read_full in reads:
        for read_length in range(10, len(read_full), 3):
            read = read_full[:read_length]
            n = len(read)
            q, i, d, g = (np.full(n, v, np.uint8) for v in (30, 45, 40, 10))
            hmm = oracle.OraclePairHMM(n, len(prefix) + len(root_1))
            hmm.do_not_use_tristate_correction()
            for prefix_start in range(len(prefix), -1, -7):
                p = prefix[prefix_start:]
                hap_1, hap_2, hap_3 = p + root_1, p + root_2, p + root_3
                hmm.compute_read_likelihood_given_haplotype_log10(hap_1, read, q, i, d, g, True, hap_2)
                actual_2 = hmm.compute_read_likelihood_given_haplotype_log10(hap_2, read, q, i, d, g, False, hap_3)
                expected_2 = hmm.compute_read_likelihood_given_haplotype_log10(hap_2, read, q, i, d, g, True, None)
                assert abs(actual_2 - expected_2) <= 1e-9


def test_transition_probabilities_closed_form():
    # pair_hmm_model_unit_tests.rs:16-19,81-138, tolerance 1e-9
    import ctypes as C
    lib = oracle.lib()
    for ins in (30, 45, 20, 10, 5, 60, 123):
        for dele in (30, 45, 20, 10, 5, 60, 123):
            for gcp in (10, 20, 5):
                dest = (C.c_double * 6)()
                lib.oracle_qual_to_trans_probs(dest, ins, dele, gcp)
                mn, mx = min(ins, dele), max(ins, dele)
                mm = 1.0 - 10.0 ** lib.oracle_approximate_log10_sum_log10(-0.1 * mn, -0.1 * mx)  # match_to_match_prob_static
                want = [mm, 1.0 - rc.q2e(gcp), rc.q2e(ins), rc.q2e(gcp), rc.q2e(dele), rc.q2e(gcp)]
                for a, e in zip(dest, want):
                    assert abs(a - e) <= 1e-9
    # table lookup vs direct formula agree for every pair incl. the >254 fallback (:442-461)
    assert lib.oracle_match_to_match_prob(0, 0) == 0.0  # table clamps 1 - min(1, p_ins + p_del) (:66-69)
    for a, b in ((6, 6), (40, 45), (254, 254), (255, 30), (255, 255)):
        direct = 1.0 - 10.0 ** lib.oracle_approximate_log10_sum_log10(-0.1 * min(a, b), -0.1 * max(a, b))
        assert abs(lib.oracle_match_to_match_prob(a, b) - direct) <= 1e-9


def test_approximate_log10_sum_log10():
    # math_utils_unit_tests.rs:35-160 family: exact on integer-phred differences, ~1e-3..1e-9 otherwise
    lib = oracle.lib()
    f = lib.oracle_approximate_log10_sum_log10
    assert f(-math.inf, -3.0) == -3.0 and f(-3.0, -math.inf) == -3.0
    assert f(-10.0, -1.0) == -1.0  # diff >= 8 -> larger term
    for a, b in ((0.0, 0.0), (-1.0, -2.0), (-0.3, -4.5), (-7.1, -0.2)):
        assert abs(f(a, b) - math.log10(10 ** a + 10 ** b)) < 1e-4
        assert f(a, b) == f(b, a)


def test_empty_read_is_minus_infinity_and_read_longer_than_haplotype_is_legal():
    hmm = oracle.OraclePairHMM(0, 5)
    e = np.zeros(0, np.uint8)
    assert hmm.compute_read_likelihood_given_haplotype_log10(b"ACGTA", b"", e, e, e, e) == -math.inf
    hmm = oracle.OraclePairHMM(12, 5)
    q = np.full(12, 30, np.uint8)
    assert hmm.compute_read_likelihood_given_haplotype_log10(b"ACGTA", b"ACGTAACGTAAC", q, q, q, q) < 0.0
