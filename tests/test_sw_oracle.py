"""oracle/sw_oracle.c (the reference's scalar Smith-Waterman arm restated in C) against the reference's own asserted
cases (tests/golden/smith_waterman_cases.json, made by tests/golden/make_sw_cases.py from
tests/smith_waterman_aligner_unit_tests.rs) and against a pure-Python restatement on random small inputs."""
import json
import os

import numpy as np

from oracle import oracle

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "smith_waterman_cases.json")))


def consumed(cigar):
    ref = sum(int(e) >> 4 for e in cigar if int(e) & 15 in (0, 2))
    alt = sum(int(e) >> 4 for e in cigar if int(e) & 15 in (0, 1, 4))
    return ref, alt


def test_asserted_cases_of_the_reference():
    assert len(GOLD["asserted"]) == 9
    for c in GOLD["asserted"]:
        cig, off = oracle.sw_align(c["reference"], c["read"], c["params"], c["strategy"])
        assert (off, oracle.cigar_to_string(cig)) == (c["expected_offset"], c["expected_cigar"]), c["source"]


def test_identical_alignments_with_differing_flank_lengths():
    """tests/smith_waterman_aligner_unit_tests.rs:320-378: the indel elements of the alignment must not depend on how
    much matching flank surrounds them."""
    f = GOLD["flank_pairs"]
    pad = "N" * 10
    p = GOLD["params"]["NEW_SW_PARAMETERS"]
    a, _ = oracle.sw_align(pad + f["padded_ref"] + pad, pad + f["padded_hap"] + pad, p, "SoftClip")
    b, _ = oracle.sw_align(pad + f["not_padded_ref"] + pad, pad + f["not_padded_hap"] + pad, p, "SoftClip")
    indels = lambda cig: [(int(e) & 15, int(e) >> 4) for e in cig if int(e) & 15 != 0]  # noqa: E731
    assert len(a) == len(b) and indels(a) == indels(b) and len(indels(a)) >= 2


def test_long_pairs_every_strategy_and_parameter_set_is_a_valid_alignment():
    """The three long pairs of test_avx_mode: the CIGAR consumes exactly the read, and (InDel / LeadingInDel: offset 0)
    stays inside the reference."""
    for pair in GOLD["avx_equals_scalar_pairs"]:
        ref, read = pair["reference"], pair["read"]
        for pname in ("NEW_SW_PARAMETERS", "STANDARD_NGS", "ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS"):
            for strategy in ("InDel", "SoftClip", "LeadingInDel", "Ignore"):
                cig, off = oracle.sw_align(ref, read, GOLD["params"][pname], strategy)
                r, a = consumed(cig)
                assert a == len(read), (pair["source"], pname, strategy)
                if strategy == "InDel":
                    assert off == 0 and r == len(ref)
                if strategy in ("SoftClip", "LeadingInDel"):
                    assert 0 <= off and off + r <= len(ref)


def _py_align(ref, alt, params, strategy):
    """Independent, deliberately naive restatement (full 2-D matrices, explicit gap scans) for small inputs: every cell
    looks at ALL gap lengths instead of the reference's running best, with the reference's tie rules (the longest
    best gap wins on `>` ... i.e. the earliest-opened among equals).  Returns the score matrix only."""
    wm, wx, wo, we = params
    n, m = len(ref), len(alt)
    sw = np.zeros((n + 1, m + 1), np.int64)
    if strategy in ("InDel", "LeadingInDel"):
        for j in range(1, m + 1):
            sw[0, j] = wo + (j - 1) * we
        for i in range(1, n + 1):
            sw[i, 0] = wo + (i - 1) * we
    low = -(2 ** 31) // 2
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            diag = sw[i - 1, j - 1] + (wm if ref[i - 1] == alt[j - 1] else wx)
            down = max([sw[i - k, j] + wo + (k - 1) * we for k in range(1, i + 1)] + [low + i * we])
            right = max([sw[i, j - k] + wo + (k - 1) * we for k in range(1, j + 1)] + [low + j * we])
            sw[i, j] = max(-100000000, max(diag, down, right))
    return sw


def test_matrix_scores_match_a_naive_all_gap_lengths_restatement():
    """The running-best-gap optimisation of calculate_matrix (:196-247) gives the same SCORE matrix as looking at every
    gap length -- checked through the alignment score the CIGAR implies."""
    rng = np.random.default_rng(5)
    alpha = "ACGT"
    for _ in range(60):
        n, m = int(rng.integers(1, 14)), int(rng.integers(1, 14))
        ref = "".join(alpha[k] for k in rng.integers(0, 3, n))
        alt = "".join(alpha[k] for k in rng.integers(0, 3, m))
        ext = int(rng.integers(1, 12))   # opening a gap costs at least as much as extending one (true of every parameter
        params = [int(rng.integers(1, 30)), -int(rng.integers(1, 30)), -(ext + int(rng.integers(0, 50))), -ext]  # set in use)
        cig, off = oracle.sw_align(ref, alt, params, "InDel")
        # score of the returned global alignment == bottom-right cell of the naive matrix
        wm, wx, wo, we = params
        i = j = 0
        score = 0
        for e in cig:
            ln, op = int(e) >> 4, int(e) & 15
            if op == 0:
                for _k in range(ln):
                    score += wm if ref[i] == alt[j] else wx
                    i += 1
                    j += 1
            elif op == 1:
                score += wo + (ln - 1) * we
                j += ln
            elif op == 2:
                score += wo + (ln - 1) * we
                i += ln
        assert (i, j) == (n, m) and off == 0
        assert score == int(_py_align(ref, alt, params, "InDel")[n, m]), (ref, alt, params, oracle.cigar_to_string(cig))


def test_empty_sequences_are_refused_like_the_reference_asserts():
    import pytest
    with pytest.raises(AssertionError):
        oracle.sw_align("", "ACGT", [3, -1, -4, -3], "SoftClip")
    with pytest.raises(AssertionError):
        oracle.sw_align("ACGT", "", [3, -1, -4, -3], "InDel")
