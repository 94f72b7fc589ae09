"""Pins the CPU oracle to the reference's own known-answer vectors.

Fixture: tests/golden/pairhmm-testdata.txt == reference tests/resources/pairhmm-testdata.txt (data:
104 lines `hap read qual ins del gcp expected`), parsed exactly as
reference tests/vector_pair_hmm_unit_tests.rs:22-50 does, gate 1e-5 absolute (:63,:90).
"""
import numpy as np

from oracle import oracle


def test_fixture_is_the_reference_fixture(kat_rows):
    assert len(kat_rows) == 104
    assert all(len(r["read"]) == len(r["qual"]) == len(r["ins"]) == len(r["dele"]) == len(r["gcp"]) for r in kat_rows)
    assert min(len(r["hap"]) for r in kat_rows) == 41 and max(len(r["hap"]) for r in kat_rows) == 164
    assert all(int(r["qual"].min()) >= 6 for r in kat_rows)  # base quals floored at 6 (:44)


def test_oracle_matches_all_104_known_answers(kat_rows):
    worst = 0.0
    for r in kat_rows:
        hmm = oracle.OraclePairHMM(len(r["read"]), len(r["hap"]))
        got = hmm.compute_read_likelihood_given_haplotype_log10(r["hap"], r["read"], r["qual"], r["ins"], r["dele"],
                                                                r["gcp"], True, None)
        assert abs(got - r["expected"]) < 1e-5, (got, r["expected"])
        worst = max(worst, abs(got - r["expected"]))
    assert worst < 7e-6  # measured 6.1e-6; the fixture itself stores 7 significant digits


def test_oracle_batch_driver_equals_single_calls(kat_rows):
    from lorikeet_amd.batch import Read, RegionBatch
    b = RegionBatch.from_regions([([Read(r["read"], r["qual"], r["ins"], r["dele"], r["gcp"])], [r["hap"]])
                                  for r in kat_rows])
    one = oracle.compute_batch(b.as_dict(), n_threads=1)
    many = oracle.compute_batch(b.as_dict(), n_threads=4)
    assert np.array_equal(one, many)
    assert np.max(np.abs(one - np.array([r["expected"] for r in kat_rows]))) < 1e-5
