"""Test cases restated from the reference's own PairHMM tests (tests/pair_hmm_unit_tests.rs).

Every generator yields dicts(hap, read, quals, ins, dele, gcp, expected, tol, kind) for the scalar path
with tristate correction DISABLED (the reference tests call do_not_use_tristate_correction(),
:116,:340,:379,...).  The same cases are run against the CPU oracle (not gpu) and the HIP path (gpu).
`kind`: "abs" = |got-expected| <= tol, "le0" = got <= 0.
"""
import math

import numpy as np

CONTEXT = b"ACGTAATGACGATTGCA"          # pair_hmm_unit_tests.rs:28
LEFT_FLANK = b"GATTTATCATCGAGTCTGC"      # :29
RIGHT_FLANK = b"CATGGATCGTTATCAGCTATCTCGAGGGATTCACTTAACAGTTTTA"  # :30
MASSIVE_QUAL = 100                       # :31
BASES = b"ACGT"


def q2e(q):
    return 10.0 ** (q / -10.0)


def _with_context(bases, left, right):  # as_bytes, :156-165
    return (LEFT_FLANK if left else b"") + CONTEXT + bases + CONTEXT + (RIGHT_FLANK if right else b"")


def _anchored(n, micro_len, qual, do_gop):  # qual_as_bytes with anchor_indel = true, :133-153
    q = np.full(n, MASSIVE_QUAL, np.uint8)
    if do_gop:
        q[len(CONTEXT)] = qual
    else:
        q[len(CONTEXT):len(CONTEXT) + micro_len] = qual
    return q


def basic_likelihood_cases(extensive=True):
    """make_basic_likelihood_tests, :169-326.  expected = -Q/10 + 0.03 + log10(1/H), tolerance 0.2."""
    base_quals = [10, 20, 30, 40, 50] if extensive else [30]
    indel_quals = [20, 30, 40, 50] if extensive else [40]
    gcps = [8, 10, 20] if extensive else [10]
    sizes = [2, 3, 4, 5, 7, 8, 9, 10, 20, 30, 35] if extensive else [2]

    def case(ref, read, bq, iq, gcp, expected_qual, left, right):
        hap = _with_context(ref, left, right)
        rd = _with_context(read, False, False)
        n = len(rd)
        return dict(hap=hap, read=rd, quals=_anchored(n, len(read), bq, False), ins=_anchored(n, len(read), iq, True),
                    dele=_anchored(n, len(read), iq, True), gcp=_anchored(n, len(read), gcp, False),
                    expected=expected_qual / -10.0 + 0.03 + math.log10(1.0 / len(hap)), tol=0.2, kind="abs")

    for bq in base_quals:
        for iq in indel_quals:
            for gcp in gcps:
                for rb in BASES:
                    for qb in BASES:
                        yield case(bytes([rb]), bytes([qb]), bq, iq, gcp, 0 if rb == qb else bq, False, False)
                for size in sizes:
                    for base in BASES:
                        expected = iq + (size - 2) * gcp
                        for insertion_p in (True, False):
                            small, big = bytes([base]), bytes([base]) * size
                            ref, read = (small, big) if insertion_p else (big, small)
                            for left, right in ((False, False), (True, False), (False, True), (True, True)):
                                yield case(ref, read, bq, iq, gcp, expected, left, right)


def mismatch_every_position_cases():
    """:328-405, relative_eq epsilon 1e-2."""
    for hap, offset, centred in ((b"TTCTCTTCTGTTGTGGCTGGTTTTCTCTTCTGTTGTGGCTGGTTTTCTCTTCTGTTGTGGCTGGTT", 2, True),
                                 (b"TTCTCTTCTGTTGTGGCTGGTT", 2, False)):
        match_qual, mismatch_qual, indel_qual = 90, 20, 80
        n = len(hap) - (2 * offset if centred else offset)
        gop = np.full(n, indel_qual, np.uint8)
        for k in range(n):
            quals = np.full(n, match_qual, np.uint8)
            quals[k] = mismatch_qual
            m_read = bytearray(hap[offset:len(hap) - offset] if centred else hap[offset:])
            m_read[k] = ord("T") if m_read[k] == ord("C") else ord("C")
            expected = math.log10((1.0 / len(hap)) * (1.0 - q2e(match_qual)) ** (n - 1) * q2e(mismatch_qual))
            yield dict(hap=hap, read=bytes(m_read), quals=quals, ins=gop, dele=gop, gcp=gop, expected=expected,
                       tol=1e-2, kind="abs")


def _expected_matching(read_len, ref_len, base_qual, ins_qual):  # :471-492
    ic = abs(ref_len - read_len + 1.0) / ref_len
    if read_len < ref_len:
        return math.log10(ic * (1.0 - q2e(base_qual)) ** read_len)
    if read_len > ref_len:
        return math.log10(ic * (1.0 - q2e(base_qual)) ** ref_len * q2e(ins_qual) ** (read_len - ref_len))
    return 0.0


def hmm_provider_cases():
    """make_hmm_provider / hmm_provider_simple, :407-539."""
    def flat(n, q):
        return np.full(n, q, np.uint8)
    for read_size in (1, 2, 5, 10):
        rb = b"A" * read_size
        # test_read_same_as_haplotype
        yield dict(hap=rb, read=rb, quals=flat(read_size, 20), ins=flat(read_size, 37), dele=flat(read_size, 37),
                   gcp=flat(read_size, 10), expected=None, tol=None, kind="le0")
        for ref_size in (1, 2, 5, 10):
            if ref_size > read_size:
                # test_multiple_read_matches_in_haplotype
                yield dict(hap=b"CC" + b"A" * ref_size + b"GGA", read=rb, quals=flat(read_size, 20),
                           ins=flat(read_size, 37), dele=flat(read_size, 37), gcp=flat(read_size, 10), expected=None,
                           tol=None, kind="le0")
                # test_all_matching_read (1e-3)
                yield dict(hap=b"A" * ref_size, read=rb, quals=flat(read_size, 20), ins=flat(read_size, 100),
                           dele=flat(read_size, 100), gcp=flat(read_size, 100),
                           expected=_expected_matching(read_size, ref_size, 20, 100), tol=1e-3, kind="abs")


def big_read_cases():
    """make_big_read_hmm_provider, :541-597: up to 800 x 2000, result must be a valid log10 prob."""
    read_1, ref_1 = b"ACCAAGTAGTCACCGT", b"ACCAAGTAGTCACCGTAACG"
    for n_read in (1, 2, 10, 20, 50):
        for n_ref in (1, 2, 10, 20, 100):
            if n_ref > n_read:
                rd, rf = read_1 * n_read, ref_1 * n_ref
                n = len(rd)
                yield dict(hap=rf, read=rd, quals=np.full(n, 30, np.uint8), ins=np.full(n, 40, np.uint8),
                           dele=np.full(n, 40, np.uint8), gcp=np.full(n, 10, np.uint8), expected=None, tol=None,
                           kind="le0")


def max_lengths_case():
    """test_max_lengths_bigger_than_provided_read, :599-635 (one real-looking read, runs must not fail)."""
    read = b"CTATCTTAGTAAGCCCCCATACCTGCAAATTTCAGGATGTCTCCTCCAAAAATCAACA"
    ref = (b"CTATCTTAGTAAGCCCCCATACCTGCAAATTTCAGGATGTCTCCTCCAAAAATCAAAACTTCTGAGAAAAAAAAAAAAAATTAAATCAAACCCTGATTCCTT"
           b"AAAGGTAGTAAAAAAACATCATTCTTTCTTAGTGGAATAGAAACTAGGTCAAAAGAACAGTGATTC")
    quals = [35, 34, 31, 32, 35, 34, 32, 31, 36, 30, 31, 32, 36, 34, 33, 32, 32, 32, 33, 32, 30, 35, 33, 35, 36, 36, 33,
             33, 33, 32, 32, 32, 37, 33, 36, 35, 33, 32, 34, 31, 36, 35, 35, 35, 35, 33, 34, 31, 31, 30, 28, 27, 26, 29,
             26, 25, 29, 29]
    ins = [46, 46, 46, 46, 46, 47, 45, 46, 45, 48, 47, 44, 45, 48, 46, 43, 43, 42, 48, 48, 45, 47, 47, 48, 48, 47, 48, 45,
           38, 47, 45, 39, 47, 48, 47, 47, 48, 46, 49, 48, 49, 48, 46, 47, 48, 44, 44, 43, 39, 32, 34, 36, 46, 48, 46, 44,
           45, 45]
    dele = [44, 44, 44, 43, 45, 44, 43, 42, 45, 46, 45, 43, 44, 47, 45, 40, 40, 40, 45, 46, 43, 45, 45, 44, 46, 46, 46,
            43, 35, 44, 43, 36, 44, 45, 46, 46, 44, 44, 47, 43, 47, 45, 45, 45, 46, 45, 45, 46, 44, 35, 35, 35, 45, 47, 45,
            44, 44, 43]
    # NB the reference passes a 58-entry quality list for a 58-base read
    n = len(read)
    assert n == len(quals) == len(ins) == len(dele)
    return dict(hap=ref, read=read, quals=np.array(quals, np.uint8), ins=np.array(ins, np.uint8),
                dele=np.array(dele, np.uint8), gcp=np.full(n, 10, np.uint8), expected=None, tol=None, kind="le0")


def haplotype_indexing_inputs():
    """make_haplotype_indexing_provider, :725-814: (prefix, roots, reads)."""
    prefix = b"AACCGGTTTTTGGGCCCAAACGTACGTACAGTTGGTCAACATCGATCAGGTTCCGGAGTAC"
    root_1, root_2, root_3 = b"ACGTGTCAAACCGGGTT", b"ACGTGTCACACTGGGTT", b"ACGTGTCACTCCGCGTT"
    reads = [b"ACGTGTCACACTGGATT", root_1, root_2, b"ACGTGTCACACTGGATTCGAT", b"CCAGTAACGTGTCACACTGGATTCGAT"]
    return prefix, (root_1, root_2, root_3), reads


def check(case, got):
    assert not math.isnan(got)
    if case["kind"] == "le0":
        assert got <= 0.0, got
    else:
        assert abs(got - case["expected"]) <= case["tol"], (got, case["expected"], case["tol"])


def to_batch(cases):
    """One region (1 read x 1 haplotype) per case."""
    from lorikeet_amd.batch import Read, RegionBatch
    return RegionBatch.from_regions([([Read(c["read"], c["quals"], c["ins"], c["dele"], c["gcp"])], [c["hap"]])
                                     for c in cases])
