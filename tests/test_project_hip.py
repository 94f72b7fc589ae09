"""phmm_project_to_reference on the MI355X (NOTEBOOK.md §15): the read -> haplotype alignment projected onto the reference, read
by read EQUAL (status, position, CIGAR) to oracle/cigar_oracle.c, which the reference's own test data pin
(tests/test_cigar_oracle.py); and the reference's create_read_aligned_to_ref cases through the device directly."""
import numpy as np
import pytest

from lorikeet_amd import realign
from lorikeet_amd.batch import RegionBatch
from lorikeet_amd.smith_waterman import ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, SmithWatermanAligner
from oracle import oracle

from project_scenarios import make_read as _read, oracle_read as _oracle_read, scenario as _scenario

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,low_complexity", [(1, False), (2, False), (3, True), (4, True), (5, False)])
def test_projection_equals_the_oracle(hip_engine, seed, low_complexity):
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(seed, low_complexity=low_complexity)
    lk = hip_engine.compute(b)
    best, aligned = realign.realign_reads_to_their_best_haplotype(hip_engine, b, lk)
    got = realign.project_to_reference(hip_engine, b, best.allele_index, aligned, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
    reg = np.repeat(np.arange(b.n_regions), np.diff(b.region_read_off.astype(np.int64)))
    n_ok = n_indel = 0
    for r in range(b.n_reads):
        st, pos, cig = _oracle_read(b, r, reg[r], best.allele_index[r], hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
        assert got.status[r] == st, (r, got.status[r], st)
        if st == 0:
            assert got.new_pos[r] == pos and oracle.cigar_to_string(got.cigars[r]) == cig, (r, oracle.cigar_to_string(got.cigars[r]), cig)
            n_ok += 1
            n_indel += ("I" in cig) or ("D" in cig)
    assert n_ok > b.n_reads // 2 and n_indel > 0


def test_every_read_against_every_haplotype(hip_engine):
    """Not only the best allele: every (read, haplotype) pair of a scenario, including hopeless ones -- statuses of the
    reference's panics included."""
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(11, n_regions=3)
    reg = np.repeat(np.arange(b.n_regions), np.diff(b.region_read_off.astype(np.int64)))
    nh = np.diff(b.region_hap_off.astype(np.int64))
    al = SmithWatermanAligner(hip_engine)
    for k in range(int(nh.max())):
        best = np.where(k < nh[reg], k, -1).astype(np.int32)
        idx = np.where(best >= 0, b.region_hap_off[:-1].astype(np.int64)[reg] + best, -1)
        haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(b.n_haps)]
        reads = [b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])] for r in range(b.n_reads)]
        aligned = al.align_indexed(haps, reads, idx, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, "SoftClip")
        got = realign.project_to_reference(hip_engine, b, best, aligned, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
        for r in range(b.n_reads):
            st, pos, cig = _oracle_read(b, r, reg[r], best[r], hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
            assert got.status[r] == st, (k, r, got.status[r], st)
            if st == 0:
                assert got.new_pos[r] == pos and oracle.cigar_to_string(got.cigars[r]) == cig, (k, r)


def _one(hip_engine, read, hap, hap_cigar, hap_start, reference, ref_start, original="10M"):
    """One read, one haplotype + the reference haplotype, through phmm_sw_align + phmm_project_to_reference."""
    u8 = lambda s: np.frombuffer(s.encode() if isinstance(s, str) else s, np.uint8)  # noqa: E731
    haps = [u8(reference), u8(hap)] if hap != reference else [u8(reference)]
    b = RegionBatch.from_regions([([_read(bytes(u8(read)))], haps)])
    k = len(haps) - 1
    aligned = SmithWatermanAligner(hip_engine).align_indexed(haps, [u8(read)], [k], ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, "SoftClip")
    cigs = [oracle.parse_cigar("%dM" % len(reference))] + ([oracle.parse_cigar(hap_cigar)] if k else [])
    if not k:
        cigs = [oracle.parse_cigar(hap_cigar)]
    got = realign.project_to_reference(hip_engine, b, [k], aligned, cigs, [0, hap_start][:k + 1] if k else [hap_start], [0], [ref_start],
                                       [oracle.parse_cigar(original)])
    assert got.status[0] == 0
    return int(got.new_pos[0]), oracle.cigar_to_string(got.cigars[0])


def test_reference_cases_of_create_read_aligned_to_ref(hip_engine):
    """tests/alignment_utils_unit_tests.rs:156-290 through the device."""
    hap = "ACTGAAGGTTCC"
    all_m = "%dM" % len(hap)
    for i in range(-1, len(hap)):
        read = bytearray(hap.encode())
        if i != -1:
            read[i] = ord("A")
        assert _one(hip_engine, bytes(read), hap, all_m, 0, hap, 10) == (10, all_m)
    for pad in range(1, 10):
        assert _one(hip_engine, "N" * pad + hap, hap, all_m, 0, hap, 10) == (10, "%dI%s" % (pad, all_m))
        assert _one(hip_engine, hap + "N" * pad, hap, all_m, 0, hap, 10) == (10, "%s%dI" % (all_m, pad))
    for ref_start in range(1, 10, 3):
        for hap_start in range(ref_start, 10 + ref_start, 3):
            assert _one(hip_engine, hap, hap, all_m, hap_start, hap, ref_start) == (ref_start + hap_start, all_m)
    reference = "GGGATCCTGCTACAAAGGTGAAACCCAGGAGAGTGTGGAGTCCAGAGTGTTGCCAGGACCCAGGCACAGGCATTAGTGCCCGTTGGAGAAAACAGGGGAATCCCGAAGAAATGGTGGGTCCTGGCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC"
    haplotype = "GGGATCCTGCTACAAAGGTGAAACCCAGGAGAGTGTGGAGTCCAGAGTGTTGCCAGGACCCAGGCACAGGCATTAGTGCCCGTTGGAGAAAACGGGAATCCCGAAGAAATGGTGGGTCCTGGCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC"
    read = "CCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC"
    assert _one(hip_engine, read, haplotype, "93M2D92M", 553, reference, 13011) == (13011 + 553 + 123, "64M")


def test_broken_inputs_fail_read_by_read_like_the_reference(hip_engine):
    """Haplotype CIGARs that do not fit their bases (too short, too long, clips and skips in odd places): the reference
    panics or returns Err for some reads (read past the end of the reference, builder errors, a cigar that does not cover
    the read ...) -- the device gives those reads the same negative status as the oracle and realigns the others alike."""
    rng = np.random.default_rng(99)
    seen = {}
    for trial in range(12):
        b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(200 + trial, n_regions=3)
        for a in range(b.n_haps):
            if rng.random() < 0.7:
                n = int(rng.integers(1, 6))
                ops = rng.choice(list("MMMMIDDNSH=X"), n)
                hap_cigars[a] = oracle.parse_cigar("".join("%d%s" % (int(rng.integers(1, 120)), o) for o in ops))
        reg = np.repeat(np.arange(b.n_regions), np.diff(b.region_read_off.astype(np.int64)))
        nh = np.diff(b.region_hap_off.astype(np.int64))
        best = rng.integers(0, nh[reg]).astype(np.int32)
        idx = b.region_hap_off[:-1].astype(np.int64)[reg] + best
        haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(b.n_haps)]
        reads = [b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])] for r in range(b.n_reads)]
        aligned = SmithWatermanAligner(hip_engine).align_indexed(haps, reads, idx, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, "SoftClip")
        got = realign.project_to_reference(hip_engine, b, best, aligned, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
        for r in range(b.n_reads):
            st, pos, cig = _oracle_read(b, r, reg[r], best[r], hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
            assert got.status[r] == st, (trial, r, got.status[r], st, oracle.cigar_to_string(hap_cigars[int(idx[r])]))
            if st == 0:
                assert got.new_pos[r] == pos and oracle.cigar_to_string(got.cigars[r]) == cig, (trial, r)
            seen[st] = seen.get(st, 0) + 1
    assert seen.get(0, 0) > 50 and sum(v for k, v in seen.items() if k < 0) > 20, seen


@pytest.mark.parametrize("seed,low_complexity,chunks", [(21, False, 0), (22, True, 0), (23, False, 3), (24, True, 5)])
def test_realign_reads_in_one_call_equals_the_three_steps(hip_engine, seed, low_complexity, chunks):
    """phmm_realign_reads == phmm_realign_to_best followed by phmm_project_to_reference (and both equal the oracle): the
    alignments stay on the device; also cut into pipelined pieces, and with output slots that are too small at first."""
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(seed, n_regions=7, low_complexity=low_complexity)
    lk = hip_engine.compute(b)
    rng = np.random.default_rng(seed)
    pri = rng.integers(-2, 2, b.n_haps).astype(np.int32)
    keep = (rng.random(b.n_reads) > 0.1).astype(np.uint8)
    best0, aligned = realign.realign_reads_to_their_best_haplotype(hip_engine, b, lk, pri, keep)
    want = realign.project_to_reference(hip_engine, b, best0.allele_index, aligned, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
    try:
        hip_engine.set_switch("sw_chunks", chunks)
        best, got = realign.realign_reads(hip_engine, b, lk, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, pri, keep, capacity=2)
    finally:
        hip_engine.set_switch("sw_chunks", 0)
    assert np.array_equal(best.allele_index, best0.allele_index) and np.array_equal(best.confidence, best0.confidence, equal_nan=True)
    assert np.array_equal(got.status, want.status) and np.array_equal(got.new_pos, want.new_pos)
    for r in range(b.n_reads):
        assert np.array_equal(got.cigars[r], want.cigars[r]), r
    reg = np.repeat(np.arange(b.n_regions), np.diff(b.region_read_off.astype(np.int64)))
    for r in range(0, b.n_reads, 3):
        st, pos, cig = _oracle_read(b, r, reg[r], best.allele_index[r], hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
        assert got.status[r] == st and (st != 0 or (got.new_pos[r] == pos and oracle.cigar_to_string(got.cigars[r]) == cig)), r


def test_realign_reads_grows_its_own_alignment_slots(hip_engine):
    """A read whose alignment to its haplotype has more CIGAR elements than the 24 the library reserves per read on the
    device: the call notices, runs again with larger slots, and the result is the oracle's."""
    rng = np.random.default_rng(5)
    unit = b"ACGTTGCAAGCT"
    hap = b"".join(unit + bytes([b"ACGT"[int(x)]]) * 3 for x in rng.integers(0, 4, 40))
    read = bytearray(hap[5:5 + 420])
    for k in range(14):   # a deletion every 30 bases: 29 elements
        del read[30 * k + 10 - 2 * k:30 * k + 12 - 2 * k]
    read = bytes(read)
    u8 = lambda s: np.frombuffer(s, np.uint8)  # noqa: E731
    b = RegionBatch.from_regions([([_read(read), _read(hap[50:130])], [u8(hap)])])
    cigs = [oracle.parse_cigar("%dM" % len(hap))]
    orig = [oracle.parse_cigar("%dM" % len(read)), oracle.parse_cigar("80M")]
    best, got = realign.realign_reads(hip_engine, b, hip_engine.compute(b), cigs, [0], [0], [100], orig)
    for r in range(2):
        st, pos, cig = _oracle_read(b, r, 0, 0, cigs, [0], [0], [100], orig)
        assert got.status[r] == st == 0 and got.new_pos[r] == pos and oracle.cigar_to_string(got.cigars[r]) == cig
    assert len(got.cigars[0]) > 24


def test_small_calls_store_results_into_the_mirror_or_copy_them_same_answer(hip_engine):
    """A call of one region fetches its inputs from the pinned mirror with a kernel and stores results and status words
    into it (no copy engine); switch `sw_no_zero_copy` takes the copies of large calls instead.  Same answer either
    way, for the fused call, the plain alignment and a call whose output slots are too small."""
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(77, n_regions=1)
    lk = hip_engine.compute(b)
    haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(b.n_haps)]
    reads = [b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])] for r in range(b.n_reads)]
    idx = np.arange(b.n_reads) % b.n_haps
    results = []
    try:
        for off in (0, 1):
            hip_engine.set_switch("sw_no_zero_copy", off)
            best, got = realign.realign_reads(hip_engine, b, lk, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
            aligned = SmithWatermanAligner(hip_engine).align_indexed(haps, reads, idx, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, "SoftClip")
            gapped = [(b"ACGTACGTAC" * 6, b"ACGTAC" + b"TTT" + b"GTACACGTAC" * 3 + b"GG" + b"ACGTACGTAC")]
            tight = SmithWatermanAligner(hip_engine).align_batch(gapped, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, "SoftClip", capacity=1)[0]
            roomy = SmithWatermanAligner(hip_engine).align_batch(gapped, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, "SoftClip")[0]
            assert len(roomy.elements) > 1 and np.array_equal(tight.elements, roomy.elements)  # (the status word said so)
            results.append((best, got, aligned))
    finally:
        hip_engine.set_switch("sw_no_zero_copy", 0)
    (b0, g0, a0), (b1, g1, a1) = results
    assert np.array_equal(b0.allele_index, b1.allele_index) and np.array_equal(b0.likelihood, b1.likelihood)
    assert np.array_equal(b0.confidence, b1.confidence, equal_nan=True)
    assert np.array_equal(g0.status, g1.status) and np.array_equal(g0.new_pos, g1.new_pos)
    for r in range(b.n_reads):
        assert np.array_equal(g0.cigars[r], g1.cigars[r]) and np.array_equal(a0[r].elements, a1[r].elements) and a0[r].alignment_offset == a1[r].alignment_offset
    reg = np.zeros(b.n_reads, np.int64)
    for r in range(b.n_reads):
        st, pos, cig = _oracle_read(b, r, reg[r], b0.allele_index[r], hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
        assert g0.status[r] == st and (st != 0 or (g0.new_pos[r] == pos and oracle.cigar_to_string(g0.cigars[r]) == cig)), r


def test_a_region_without_haplotypes_leaves_its_reads_unchanged(hip_engine):
    """a region with reads but no haplotypes (nothing assembled) has no best alleles; phmm_realign_reads and
    phmm_region_compute leave its reads as they are instead of refusing the whole call for its missing reference haplotype."""
    from lorikeet_amd import _lib, region
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(91, n_regions=3)
    regions = []
    for g in range(3):
        reads = [_read(bytes(b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])])) for r in range(int(b.region_read_off[g]), int(b.region_read_off[g + 1]))]
        haps = [] if g == 1 else [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(int(b.region_hap_off[g]), int(b.region_hap_off[g + 1]))]
        regions.append((reads, haps))
    keep_h = [a for g in (0, 2) for a in range(int(b.region_hap_off[g]), int(b.region_hap_off[g + 1]))]
    b2 = RegionBatch.from_regions(regions)
    hc, hs = [hap_cigars[a] for a in keep_h], [hap_starts[a] for a in keep_h]
    rh = [ref_hap[0], -1, ref_hap[2]]
    lk = hip_engine.compute(b2)
    best, got = realign.realign_reads(hip_engine, b2, lk, hc, hs, rh, ref_start, orig_cigars)
    mid = slice(int(b2.region_read_off[1]), int(b2.region_read_off[2]))
    assert np.all(best.allele_index[mid] == -1) and np.all(got.status[mid] == _lib.PHMM_PROJECT_UNCHANGED)
    assert np.any(got.status[:mid.start] == 0) and np.any(got.status[mid.stop:] == 0)
    cfg = _lib.EngineConfig()
    cfg.constant_gcp, cfg.base_quality_score_threshold, cfg.symmetrically_normalize_alleles_to_reference = 10, 18, 1
    cfg.log10_global_read_mismapping_rate, cfg.read_disqualification_scale, cfg.expected_error_rate_per_base = -4.5, 1.0, 0.02
    fused = region.region_compute(hip_engine, cfg, b2, np.full(b2.n_reads, 60, np.uint8), hc, hs, rh, ref_start, orig_cigars)
    assert np.all(fused.best.allele_index[mid] == -1) and np.all(fused.reads.status[mid] == _lib.PHMM_PROJECT_UNCHANGED)
    assert np.any(fused.reads.status[:mid.start] == 0)


def test_plain_reads_take_the_short_way_to_the_same_answer(hip_engine):
    """A read aligned to its haplotype as one M element over its whole length, the haplotype one M element against the
    reference: the device skips the builders (project_read, `plain`).  Haplotype CIGARs of one M element that is shorter than,
    as long as and longer than the haplotype's bases (the reference pads with 1000M either way), every haplotype of the
    region: status, position and CIGAR equal to the oracle's general procedure."""
    rng = np.random.default_rng(5)
    n_plain = 0
    for trial, m_len in enumerate([1, 17, 150, None, 5000]):
        b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(300 + trial, n_regions=2)
        for a in range(b.n_haps):
            hap_len = int(b.hap_off[a + 1] - b.hap_off[a])
            hap_cigars[a] = oracle.parse_cigar("%dM" % (hap_len if m_len is None else m_len))
        reg = np.repeat(np.arange(b.n_regions), np.diff(b.region_read_off.astype(np.int64)))
        nh = np.diff(b.region_hap_off.astype(np.int64))
        best = rng.integers(0, nh[reg]).astype(np.int32)
        idx = b.region_hap_off[:-1].astype(np.int64)[reg] + best
        haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(b.n_haps)]
        reads = [b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])] for r in range(b.n_reads)]
        aligned = SmithWatermanAligner(hip_engine).align_indexed(haps, reads, idx, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, "SoftClip")
        got = realign.project_to_reference(hip_engine, b, best, aligned, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
        for r in range(b.n_reads):
            st, pos, cig = _oracle_read(b, r, reg[r], best[r], hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
            assert got.status[r] == st, (trial, r, got.status[r], st)
            if st == 0:
                assert got.new_pos[r] == pos and oracle.cigar_to_string(got.cigars[r]) == cig, (trial, r, oracle.cigar_to_string(got.cigars[r]), cig)
                n_plain += len(aligned[r].elements) == 1
    assert n_plain > 20
