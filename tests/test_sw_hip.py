"""Smith-Waterman on the MI355X (phmm_sw_align) against the oracle (the reference's scalar arm in C)
and the reference's own asserted cases: integer work, so CIGAR and offset must be EQUAL."""
import json
import os

import numpy as np
import pytest

from lorikeet_amd import HipPairHMMEngine, PhmmError, _lib
from lorikeet_amd.smith_waterman import (ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, NEW_SW_PARAMETERS, ORIGINAL_DEFAULT,
                                         STANDARD_NGS, OverhangStrategy, Parameters, SmithWatermanAligner)
from oracle import oracle

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "smith_waterman_cases.json")))
STRATEGIES = ("SoftClip", "InDel", "LeadingInDel", "Ignore")


@pytest.fixture(scope="module")
def aligner(hip_engine):
    return SmithWatermanAligner(hip_engine)


def _same(got, ref, alt, params, strategy, ctx=None):
    cig, off = oracle.sw_align(ref, alt, [params.match_value, params.mismatch_penalty, params.gap_open_penalty,
                                          params.gap_extend_penalty], strategy)
    assert got.alignment_offset == off and np.array_equal(got.elements, cig), \
        (ctx, strategy, got, oracle.cigar_to_string(cig), off)


def test_asserted_cases_of_the_reference(aligner):
    for c in GOLD["asserted"]:
        r = aligner.align(c["reference"], c["read"], Parameters(*c["params"]), c["strategy"])
        assert (r.get_alignment_offset(), r.cigar_string()) == (c["expected_offset"], c["expected_cigar"]), c["source"]


def test_long_pairs_of_the_reference_equal_the_scalar_arm(aligner):
    """tests/smith_waterman_aligner_unit_tests.rs:999-1103: three parameter sets x the overhang strategies on the three
    long pairs (reads of 1 010 - 1 295 bases: twenty strips of 64 columns), plus the flank-length pairs (:320-378)."""
    f = GOLD["flank_pairs"]
    pad = "N" * 10
    pairs = [(p["reference"], p["read"]) for p in GOLD["avx_equals_scalar_pairs"]]
    pairs += [(pad + f["padded_ref"] + pad, pad + f["padded_hap"] + pad), (pad + f["not_padded_ref"] + pad, pad + f["not_padded_hap"] + pad)]
    pairs += [(b, a) for a, b in pairs[:2]]   # and the other way round: a long reference, 5 strips
    for params in (NEW_SW_PARAMETERS, STANDARD_NGS, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS):
        for strategy in STRATEGIES:
            for k, (got, (ref, alt)) in enumerate(zip(aligner.align_batch(pairs, params, strategy, capacity=400), pairs)):
                _same(got, ref, alt, params, strategy, k)


def _mutate(rng, seq, p_sub, p_indel):
    out = []
    alpha = b"ACGT"
    i = 0
    while i < len(seq):
        u = rng.random()
        if u < p_indel / 2:
            i += int(rng.integers(1, 8))           # deletion
        elif u < p_indel:
            out.extend(alpha[int(k)] for k in rng.integers(0, 4, int(rng.integers(1, 8))))  # insertion
        elif u < p_indel + p_sub:
            out.append(alpha[int(rng.integers(0, 4))])
            i += 1
        else:
            out.append(seq[i])
            i += 1
    return bytes(out) or b"A"


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_random_batches_equal_the_oracle(aligner, strategy):
    """Ragged batches: 1 ... 700 bases on either side (strip boundaries at 64, 128, ...), related and unrelated sequences,
    parameter sets where ties are frequent (small integers), read -> haplotype and haplotype -> reference shapes."""
    rng = np.random.default_rng({"SoftClip": 1, "InDel": 2, "LeadingInDel": 3, "Ignore": 4}[strategy])
    alpha = b"ACGT"
    pairs = []
    for k in range(260):
        n = int(rng.choice([1, 2, 5, 63, 64, 65, 127, 128, 129, 200, 333, 512, 700])) if k % 3 == 0 else int(rng.integers(1, 420))
        ref = bytes(alpha[int(x)] for x in rng.integers(0, 4 if k % 5 else 2, n))
        kind = k % 4
        if kind == 0:   # unrelated
            alt = bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 300))))
        elif kind == 1:  # a read out of the reference, with errors
            s = int(rng.integers(0, n))
            alt = _mutate(rng, ref[s:s + int(rng.integers(1, 160))], 0.03, 0.02)
        elif kind == 2:  # a haplotype: the whole reference with variants
            alt = _mutate(rng, ref, 0.01, 0.01)
        else:            # overhanging on both sides
            alt = bytes(alpha[int(x)] for x in rng.integers(0, 4, 7)) + _mutate(rng, ref[: max(1, n // 2)], 0.02, 0.02) + b"GGGTT"
        pairs.append((ref, alt))
    for params in (ORIGINAL_DEFAULT, NEW_SW_PARAMETERS, Parameters(1, -1, -1, -1), Parameters(2, -3, -5, -2)):
        got = aligner.align_batch(pairs, params, strategy)
        for k, (g, (ref, alt)) in enumerate(zip(got, pairs)):
            _same(g, ref, alt, params, strategy, k)


def test_long_sequences_one_alignment_per_wave(aligner):
    """Sequences of a few thousand bases: the four-alignments-per-wave layout no longer fits a block's LDS, every wave
    takes one alignment (several strips of 512 columns, strip edges through LDS); beyond ~8 000 bases the call is refused."""
    rng = np.random.default_rng(9)
    alpha = b"ACGT"
    ref = bytes(alpha[int(x)] for x in rng.integers(0, 4, 3100))
    pairs = [(ref, _mutate(rng, ref[200:2900], 0.01, 0.004)), (ref[:2500], _mutate(rng, ref[:2500], 0.02, 0.01)), (b"ACGT" * 30, b"ACGTTACG")]
    for strategy in ("SoftClip", "InDel"):
        for g, (r, a) in zip(aligner.align_batch(pairs, NEW_SW_PARAMETERS, strategy, capacity=64), pairs):
            _same(g, r, a, NEW_SW_PARAMETERS, strategy)
    # ... and beyond ~8 000 bases, where the bottom row and the strip edges no longer fit LDS next to the sequences and move
    # to device memory: the reference aligns any lengths (smith_waterman_aligner.rs:47-107), so does the library
    # (20 000 x 20 000 used to be refused).  One pair, every strip of it.
    big_ref = bytes(alpha[int(x)] for x in rng.integers(0, 4, 20000))
    big_alt = _mutate(rng, big_ref[300:19800], 0.004, 0.001) + bytes(alpha[int(x)] for x in rng.integers(0, 4, 200))
    for strategy, prm in (("SoftClip", ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS), ("InDel", NEW_SW_PARAMETERS)):
        g = aligner.align_batch([(big_ref, big_alt)], prm, strategy, capacity=4096)[0]
        _same(g, big_ref, big_alt, prm, strategy)
        assert len(g.elements) > 20
    with pytest.raises(PhmmError, match="too long"):
        aligner.align(b"A" * 200000, b"C" * 1000, NEW_SW_PARAMETERS, "InDel")


def test_capacity_is_reported_not_overrun(hip_engine):
    """A CIGAR that needs more elements than its slot: PHMM_ERR_CIGAR_CAPACITY, n_cigar holds the size, the guard word
    behind the slot is untouched and the other alignments of the call are valid."""
    import ctypes as C
    p = GOLD["avx_equals_scalar_pairs"][0]
    refs = [np.frombuffer(p["reference"].encode(), np.uint8), np.frombuffer(b"AAACCCCC", np.uint8)]
    alts = [np.frombuffer(p["read"].encode(), np.uint8), np.frombuffer(b"CCCCC", np.uint8)]
    ref_off = np.array([0, len(refs[0]), len(refs[0]) + 8], np.uint32)
    alt_off = np.array([0, len(alts[0]), len(alts[0]) + 5], np.uint32)
    cig_off = np.array([0, 4, 8], np.uint64)
    cigar = np.full(9, 0xdeadbeef, np.uint32)
    n_cig = np.zeros(2, np.uint32)
    off = np.zeros(2, np.int32)
    prm = NEW_SW_PARAMETERS.as_struct()
    rb, ab = np.concatenate(refs), np.concatenate(alts)
    code = hip_engine.lib.phmm_sw_align(hip_engine._h, 2, ref_off.ctypes.data_as(_lib.u32p), rb.ctypes.data_as(_lib.u8p),
                                        alt_off.ctypes.data_as(_lib.u32p), ab.ctypes.data_as(_lib.u8p), C.byref(prm),
                                        OverhangStrategy.InDel, cig_off.ctypes.data_as(_lib.u64p), cigar.ctypes.data_as(_lib.u32p),
                                        n_cig.ctypes.data_as(_lib.u32p), off.ctypes.data_as(C.POINTER(C.c_int32)))
    want, _ = oracle.sw_align(p["reference"], p["read"], [200, -150, -260, -11], "InDel")
    assert code == _lib.PHMM_ERR_CIGAR_CAPACITY and n_cig[0] == len(want) > 4
    assert n_cig[1] == 2 and oracle.cigar_to_string(cigar[4:6]) == "3D5M" and cigar[8] == 0xdeadbeef
    # the Python mirror retries with the reported sizes
    r = SmithWatermanAligner(hip_engine).align(p["reference"], p["read"], NEW_SW_PARAMETERS, "InDel")
    assert np.array_equal(r.elements, want)


def test_argument_errors(hip_engine, aligner):
    with pytest.raises(AssertionError, match="non-empty"):
        aligner.align(b"", b"ACGT", ORIGINAL_DEFAULT, "SoftClip")
    import ctypes as C
    z = np.zeros(2, np.uint32)
    one = np.array([0, 1], np.uint32)
    seq = np.frombuffer(b"A", np.uint8)
    prm = ORIGINAL_DEFAULT.as_struct()
    args = lambda ro, ao, st: (hip_engine._h, 1, ro.ctypes.data_as(_lib.u32p), seq.ctypes.data_as(_lib.u8p),  # noqa: E731
                               ao.ctypes.data_as(_lib.u32p), seq.ctypes.data_as(_lib.u8p), C.byref(prm), st,
                               np.array([0, 4], np.uint64).ctypes.data_as(_lib.u64p), np.zeros(4, np.uint32).ctypes.data_as(_lib.u32p),
                               np.zeros(1, np.uint32).ctypes.data_as(_lib.u32p), np.zeros(1, np.int32).ctypes.data_as(C.POINTER(C.c_int32)))
    assert hip_engine.lib.phmm_sw_align(*args(z, one, 0)) == _lib.PHMM_ERR_INVALID_ARG and "non-empty" in hip_engine.last_error()
    assert hip_engine.lib.phmm_sw_align(*args(one, one, 7)) == _lib.PHMM_ERR_INVALID_ARG
    assert hip_engine.lib.phmm_sw_align(*args(one, one, 1)) == _lib.PHMM_OK


def test_long_gaps_are_recovered_from_the_open_flags(aligner):
    """The kernel stores which candidate won and where gaps open, not gap lengths: a backtrack step over a long gap walks
    the flags back to the cell the gap opened at -- across the K columns of a lane, across lanes and across strips of
    512 columns (an insertion of 700 bases), and up a column for deletions of hundreds of rows."""
    rng = np.random.default_rng(77)
    alpha = b"ACGT"
    rnd = lambda n: bytes(alpha[int(x)] for x in rng.integers(0, 4, n))  # noqa: E731
    core = rnd(900)
    pairs = [
        (core[:300] + core[420:], core),                                   # 120 bases missing from the reference: insertion
        (core, core[:250] + core[610:]),                                   # 360 rows skipped: deletion
        (core[:200] + core[200:260] * 1 + core[260:], core[:200] + rnd(700) + core[200:]),   # insertion longer than a strip
        (core[:100] + rnd(333) + core[100:500], core[:500]),               # deletion of 333
        (rnd(40) + core[:150], core[:150] + rnd(35)),                      # overhangs on both sides
        (b"A" * 300, b"A" * 120 + b"C" * 90 + b"A" * 100),                 # low complexity: ties everywhere
    ]
    for params in (NEW_SW_PARAMETERS, Parameters(3, -2, -4, -1), Parameters(1, -3, -2, 0), Parameters(20000, -15000, -26000, -1100)):
        for strategy in STRATEGIES:
            for k, (g, (r, a)) in enumerate(zip(aligner.align_batch(pairs, params, strategy, capacity=256), pairs)):
                _same(g, r, a, params, strategy, k)


def test_pipelined_pieces_equal_one_piece(hip_engine, aligner):
    """Large calls are cut into pieces whose staging overlaps the previous piece's kernel (switch `sw_chunks`; by size
    otherwise): same results whatever the cut, ragged pieces and pieces of a single alignment included."""
    rng = np.random.default_rng(5)
    alpha = b"ACGT"
    pairs = []
    for k in range(203):
        ref = bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(20, 400))))
        alt = _mutate(rng, ref[int(rng.integers(0, 10)):], 0.03, 0.02) if k % 3 else bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 200))))
        pairs.append((ref, alt))
    base = None
    try:
        for lanes in (8, 16, 32, 64):   # lanes per alignment: chosen by batch size and sequence length -- forced here
            hip_engine.set_switch("sw_lanes", lanes)
            for chunks in (1, 2, 3, 7):
                hip_engine.set_switch("sw_chunks", chunks)
                for n in (len(pairs), 3):
                    got = aligner.align_batch(pairs[:n], STANDARD_NGS, "SoftClip")
                    if base is None:
                        base = got
                        for g, (r, a) in zip(got, pairs):
                            _same(g, r, a, STANDARD_NGS, "SoftClip")
                    for g, b in zip(got, base):
                        assert g.alignment_offset == b.alignment_offset and np.array_equal(g.elements, b.elements), (lanes, chunks)
    finally:
        hip_engine.set_switch("sw_chunks", 0)
        hip_engine.set_switch("sw_lanes", 0)


def test_weights_of_a_million_take_the_wide_instance_with_the_reference_clamp(aligner):
    """Scores travel times four with a tag in the low bits and without the reference's clamp at -1e8: exact while
    |weight| x (ref + alt) < 1e8.  Beyond that (used to be refused) the wide instance carries the scores as they
    are, picks the candidate by the reference's own comparisons and applies its clamp -- equal to the oracle, also where the
    clamp acts.  Where the reference's own 32-bit sums overflow (|weight| x (ref + alt) >= 1e9) the call is refused."""
    rng = np.random.default_rng(31)
    alpha = b"ACGT"
    pairs = []
    for _ in range(24):
        ref = bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(40, 260))))
        alt = _mutate(rng, ref[int(rng.integers(0, 20)):len(ref) - int(rng.integers(0, 20))], 0.05, 0.03) if rng.random() < 0.7 else \
            bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(30, 200))))
        pairs.append((ref, alt or b"A"))
    for prm in (Parameters(1000000, -1000000, -2000000, -500000), Parameters(700000, -1500000, -900000, -300000),
                Parameters(3, -2000000, -1900000, -1000000)):   # the last: unrelated pairs run into the clamp at -1e8
        for strategy in ("SoftClip", "InDel", "LeadingInDel", "Ignore"):
            for g, (r, a) in zip(aligner.align_batch(pairs, prm, strategy, capacity=64), pairs):
                _same(g, r, a, prm, strategy)
    with pytest.raises(PhmmError, match="parameters too large"):
        aligner.align(b"ACGT" * 200, b"ACGT" * 200, Parameters(1000000, -1000000, -2000000, -500000), "SoftClip")


def test_small_calls_sweep_along_the_alternate_sequence_same_alignments(hip_engine, aligner):
    """A small call whose references are longer than its alternates (at most 512 rows) runs the same matrix with the
    reference's rows shared out over the lanes and the sweep along the alternate (switch `sw_transpose`: 1 = whenever
    possible, 0 = never, -1 = by cost): same CIGARs and offsets as the oracle either way, every strategy, ragged pairs,
    references shorter than their alternates and alternates that would need several strips the other way round."""
    rng = np.random.default_rng(41)
    alpha = b"ACGT"
    pairs = []
    for k in range(150):
        ref = bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 500))))
        kind = k % 5
        if kind == 0:
            alt = bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 700))))       # unrelated, any length
        elif kind == 1:
            alt = _mutate(rng, ref, 0.02, 0.03)                                                       # haplotype-like
        elif kind == 2:
            alt = _mutate(rng, b"ACGTTG" * 3 + ref[len(ref) // 3:] + b"TTGCA" * 2, 0.03, 0.02)       # overhanging both ends
        else:
            s = int(rng.integers(0, max(1, len(ref) - 20)))
            alt = _mutate(rng, ref[s:s + int(rng.integers(20, 200))], 0.03, 0.03) or b"A"            # read-like
        pairs.append((ref, alt))
    pairs += [(b"A" * 300, b"A" * 150), (b"AC" * 200, b"CA" * 80), (b"ACGT" * 100, b"T")]
    try:
        hip_engine.set_switch("sw_lanes", 64)
        for params in (ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, Parameters(3, -2, -4, -1), Parameters(1, -3, -2, 0)):
            for strategy in STRATEGIES:
                got = {}
                for mode in (1, 0):
                    hip_engine.set_switch("sw_transpose", mode)
                    got[mode] = aligner.align_batch(pairs, params, strategy, capacity=128)
                for k, ((r, a), g1, g0) in enumerate(zip(pairs, got[1], got[0])):
                    _same(g1, r, a, params, strategy, k)
                    assert g1.alignment_offset == g0.alignment_offset and np.array_equal(g1.elements, g0.elements), (strategy, k)
    finally:
        hip_engine.set_switch("sw_transpose", -1)
        hip_engine.set_switch("sw_lanes", 0)


def test_only_sequences_that_are_aligned_must_be_non_empty(hip_engine):
    """the reference asserts on the sequences it aligns (smith_waterman_aligner.rs:65-68), not on a haplotype no
    read is aligned to, nor on the read of a skipped alignment."""
    al = SmithWatermanAligner(hip_engine)
    refs = [b"ACGTACGTTTGACCA", b"", b"TTGACCAGGA"]
    alts = [b"ACGTTTGA", b"", b"GACCAG", b"CCAGG"]
    got = al.align_indexed(refs, alts, [0, -1, 2, 2], NEW_SW_PARAMETERS, "SoftClip")
    assert got[1] is None
    for k, r in ((0, 0), (2, 2), (3, 2)):
        cig, off = oracle.sw_align(refs[r], alts[k], [200, -150, -260, -11], "SoftClip")
        assert got[k].alignment_offset == off and np.array_equal(got[k].elements, cig)
    with pytest.raises(AssertionError, match="non-empty"):
        al.align_indexed(refs, alts, [0, -1, 1, 2], NEW_SW_PARAMETERS, "SoftClip")   # the empty reference IS used


def test_tags_only_first_pass_gives_the_same_alignments(hip_engine, aligner):
    """SoftClip / Ignore calls sweep with the two-bit candidate tags only and send the alignments whose walk meets a gap
    through the full instance again (switch `sw_lite`: 1 always, 0 never, -1 where it has been paying): the same CIGARs and
    offsets either way -- reads without indels, reads with many, unrelated sequences, every lanes-per-alignment value, pieces
    -- and the counter says how many went round again."""
    rng = np.random.default_rng(17)
    alpha = b"ACGT"
    pairs = []
    for k in range(240):
        ref = bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(30, 420))))
        s = int(rng.integers(0, len(ref) // 2))
        if k % 4 == 0:      # a clean read (mismatches only): no gap on its path
            alt = _mutate(rng, ref[s:s + 150], 0.0, 0.02)
        elif k % 4 == 1:    # 3-6 % indels
            alt = _mutate(rng, ref[s:s + 150], float(rng.uniform(0.03, 0.06)), 0.01)
        elif k % 4 == 2:    # overhanging
            alt = b"TTGCA" + _mutate(rng, ref[s:s + 90], 0.01, 0.01) + b"GGGTT"
        else:               # unrelated
            alt = bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 200))))
        pairs.append((ref, alt))
    try:
        for strategy in ("SoftClip", "Ignore"):
            for params in (STANDARD_NGS, Parameters(1, -1, -1, -1)):
                hip_engine.set_switch("sw_lite", 0)
                want = aligner.align_batch(pairs, params, strategy)
                assert hip_engine.stat("sw_second_pass") == 0
                for g, (r, a) in zip(want, pairs):
                    _same(g, r, a, params, strategy)
                hip_engine.set_switch("sw_lite", 1)
                for lanes in (0, 8, 16, 32, 64):
                    hip_engine.set_switch("sw_lanes", lanes)
                    for chunks in (0, 3):
                        hip_engine.set_switch("sw_chunks", chunks)
                        got = aligner.align_batch(pairs, params, strategy)
                        again = hip_engine.stat("sw_second_pass")
                        assert 40 <= again < len(pairs), again      # the indel reads at least, the clean ones never
                        for g, b in zip(got, want):
                            assert g.alignment_offset == b.alignment_offset and np.array_equal(g.elements, b.elements), (strategy, lanes, chunks)
        # a large batch: the second pass of a short list (<= 1 024 alignments) runs in the small-call geometry, a longer
        # one in the batch's own -- 4 800 alignments with 60 and with 2 400 gapped ones
        clean = [p for k, p in enumerate(pairs) if k % 4 == 0 and len(p[1]) <= 150]
        gappy = [p for k, p in enumerate(pairs) if k % 4 == 1]
        for big in (clean * 79 + gappy, (clean[:40] + gappy[:40]) * 60):
            big = big[:4800]
            hip_engine.set_switch("sw_lite", 0)
            want = aligner.align_batch(big, STANDARD_NGS, "SoftClip")
            hip_engine.set_switch("sw_lite", 1)
            for chunks in (1, 0):
                hip_engine.set_switch("sw_chunks", chunks)
                got = aligner.align_batch(big, STANDARD_NGS, "SoftClip")
                assert hip_engine.stat("sw_second_pass") >= 40
                for g, b in zip(got, want):
                    assert g.alignment_offset == b.alignment_offset and np.array_equal(g.elements, b.elements), chunks
        # a call in pieces takes one second pass behind its last piece and patches the results; more than 4 096 alignments
        # to redo are fetched wholesale instead
        many = (gappy * 101)[:6000]
        hip_engine.set_switch("sw_lite", 0)
        hip_engine.set_switch("sw_chunks", 3)
        want = aligner.align_batch(many, STANDARD_NGS, "SoftClip")
        hip_engine.set_switch("sw_lite", 1)
        got = aligner.align_batch(many, STANDARD_NGS, "SoftClip")
        assert hip_engine.stat("sw_second_pass") > 4096
        for g, b in zip(got, want):
            assert g.alignment_offset == b.alignment_offset and np.array_equal(g.elements, b.elements)
        hip_engine.set_switch("sw_chunks", 0)
        # the default: two passes until a call meets gaps in more than three alignments of ten, then fifteen calls without
        hip_engine.set_switch("sw_lite", -1)
        hip_engine.set_switch("sw_lanes", 0)
        hip_engine.set_switch("sw_chunks", 0)
        first = aligner.align_batch(gappy, STANDARD_NGS, "SoftClip")
        assert hip_engine.stat("sw_second_pass") > len(gappy) * 0.3
        second = aligner.align_batch(gappy, STANDARD_NGS, "SoftClip")
        assert hip_engine.stat("sw_second_pass") == 0               # straight to the full instance
        for g, b in zip(first, second):
            assert g.alignment_offset == b.alignment_offset and np.array_equal(g.elements, b.elements)
        # InDel strategies never take the first pass
        hip_engine.set_switch("sw_lite", 1)
        aligner.align_batch(pairs[:20], NEW_SW_PARAMETERS, "InDel")
        assert hip_engine.stat("sw_second_pass") == 0
    finally:
        hip_engine.set_switch("sw_lite", -1)
        hip_engine.set_switch("sw_chunks", 0)
        hip_engine.set_switch("sw_lanes", 0)
