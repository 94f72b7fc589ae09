"""phmm_calculate_cigar on the MI355X: CigarUtils::calculate_cigar (src/reads/cigar_utils.rs:358-457) -- the reference's own
114 cases (tests/cigar_utils_unit_tests.rs:34-283, tests/golden/calculate_cigar_cases.json) through the device, and random
haplotype / reference pairs EQUAL to oracle_calculate_cigar."""
import json
import os

import numpy as np
import pytest

from lorikeet_amd.smith_waterman import NEW_SW_PARAMETERS, STANDARD_NGS, calculate_cigar
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "calculate_cigar_cases.json")))


def test_reference_cases_through_the_device(hip_engine):
    pairs = [(c["reference"], c["alternate"]) for c in GOLD["cases"]]
    got = calculate_cigar(hip_engine, pairs, NEW_SW_PARAMETERS, GOLD["strategy"])
    for c, g in zip(GOLD["cases"], got):
        assert g is not None and oracle.cigar_to_string(g) == c["expected_cigar"], (c, None if g is None else oracle.cigar_to_string(g))


def _variant(rng, ref):
    out, i = bytearray(), 0
    while i < len(ref):
        u = rng.random()
        if u < 0.02:
            i += int(rng.integers(1, 12))
        elif u < 0.04:
            k = int(rng.integers(1, 12))
            out += bytes(out[-k:]) if rng.random() < 0.5 and len(out) >= k else bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, k))
        elif u < 0.06:
            out.append(b"ACGT"[int(rng.integers(0, 4))])
            i += 1
        else:
            out.append(ref[i])
            i += 1
    return bytes(out)


@pytest.mark.parametrize("strategy", ["InDel", "SoftClip", "LeadingInDel", "Ignore"])
def test_random_haplotypes_equal_the_oracle(hip_engine, strategy):
    rng = np.random.default_rng({"InDel": 1, "SoftClip": 2, "LeadingInDel": 3, "Ignore": 4}[strategy])
    pairs = []
    for k in range(160):
        alphabet = 2 if k % 4 == 0 else 4   # low complexity: repeats that left-align, deletions that reach the ends
        ref = bytes(b"ACGT"[int(x)] for x in rng.integers(0, alphabet, int(rng.integers(1, 400))))
        alt = _variant(rng, ref) if k % 7 else bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(0, 60))))
        if k % 11 == 0:
            alt = ref[int(rng.integers(0, 5)):len(ref) - int(rng.integers(0, 5))]   # the haplotype lacks the ends: leading / trailing deletions
        pairs.append((ref, alt))
    pairs += [(b"", b"ACGT"), (b"ACGT", b""), (b"A", b"A"), (b"ACGTACGT", b"ACGAACGA"), (b"ACGTACGT", b"TCGAACGA")]
    for params in (NEW_SW_PARAMETERS, STANDARD_NGS):
        got = calculate_cigar(hip_engine, pairs, params, strategy, capacity=4)
        none = 0
        for (ref, alt), g in zip(pairs, got):
            want = oracle.calculate_cigar(ref, alt, [params.match_value, params.mismatch_penalty, params.gap_open_penalty, params.gap_extend_penalty], strategy)
            assert (g is None and want is None) or (g is not None and oracle.cigar_to_string(g) == want), (ref, alt, want)
            none += want is None
        assert none < len(pairs)
