"""The arithmetic of realign_reads_to_their_best_haplotype on the MI355X (SURVEY.md 8 rows f1 -> f4 joined):
phmm_best_alleles against oracle_best_alleles (indices, likelihoods and confidences EQUAL: the kernel copies and
subtracts the same doubles), phmm_sw_align_indexed against the plain call, and phmm_realign_to_best against both."""
import numpy as np
import pytest

from lorikeet_amd import PhmmError, _lib, realign, synthetic
from lorikeet_amd.batch import RegionBatch
from lorikeet_amd.smith_waterman import (ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, NEW_SW_PARAMETERS, STANDARD_NGS,
                                         SmithWatermanAligner)
from oracle import oracle

pytestmark = pytest.mark.gpu


def _structure(rng, n_regions, max_reads=40, max_haps=9, allow_empty=True):
    nr = rng.integers(0 if allow_empty else 1, max_reads + 1, n_regions)
    nh = rng.integers(0 if allow_empty else 1, max_haps + 1, n_regions)
    rro = np.concatenate([[0], np.cumsum(nr)]).astype(np.uint32)
    rho = np.concatenate([[0], np.cumsum(nh)]).astype(np.uint32)
    gap = rng.integers(0, 3, n_regions)  # out_off may leave room between the matrices
    oo = np.concatenate([[0], np.cumsum(nr * nh + gap)]).astype(np.uint64)
    return nr, nh, rro, rho, oo


class _B:  # the five arrays phmm_best_alleles reads of a batch
    def __init__(self, rro, rho, oo):
        self.region_read_off, self.region_hap_off, self.out_off = rro, rho, oo
        self.n_regions, self.n_reads = len(rro) - 1, int(rro[-1])


def _oracle_best(nr, nh, rro, rho, oo, lk, pri, keep, thr):
    n = int(rro[-1])
    best, olk, conf = np.full(n, -1, np.int32), np.full(n, -np.inf), np.full(n, np.nan)
    for g in range(len(nr)):
        if nr[g] == 0 or nh[g] == 0:
            continue
        m = lk[int(oo[g]):int(oo[g]) + nr[g] * nh[g]].reshape(nr[g], nh[g])
        b, l, c = oracle.best_alleles(m.T, None if pri is None else pri[rho[g]:rho[g + 1]], thr)
        s = slice(int(rro[g]), int(rro[g + 1]))
        best[s], olk[s], conf[s] = b, l, c
    if keep is not None:
        best[keep == 0], olk[keep == 0], conf[keep == 0] = -1, -np.inf, np.nan
    return best, olk, conf


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_best_alleles_equal_the_oracle(hip_engine, seed):
    """Random matrices on a 0.05 grid (ties and near-ties within the 0.2 threshold everywhere), random priorities with
    repeats, removed evidence, regions without reads or without alleles, gaps in out_off."""
    rng = np.random.default_rng(seed)
    nr, nh, rro, rho, oo = _structure(rng, 60)
    lk = np.round(rng.normal(-3.0, 0.4, int(oo[-1])) * 20) / 20
    lk[rng.random(len(lk)) < 0.02] = -np.inf
    pri = rng.integers(-3, 3, int(rho[-1])).astype(np.int32)
    keep = (rng.random(int(rro[-1])) > 0.1).astype(np.uint8)
    for p, k in ((pri, keep), (None, None), (pri, None)):
        got = realign.best_alleles_breaking_ties(hip_engine, _B(rro, rho, oo), lk, p, k)
        wb, wl, wc = _oracle_best(nr, nh, rro, rho, oo, lk, p, k, 0.2)
        assert np.array_equal(got.allele_index, wb)
        assert np.array_equal(got.likelihood, wl) and np.array_equal(got.confidence, wc, equal_nan=True)


def test_best_alleles_reference_property(hip_engine):
    """tests/allele_likelihoods_unit_tests.rs:250-365 through the device: the reference allele (priority 1) takes over
    whenever it is within 0.2 of the arg-max."""
    rng = np.random.default_rng(9)
    nr, nh, rro, rho, oo = _structure(rng, 40, allow_empty=False)
    oo = np.concatenate([[0], np.cumsum(nr * nh)]).astype(np.uint64)
    lk = rng.normal(0.0, 1.0, int(oo[-1]))
    ref = rng.integers(0, nh)  # the reference allele of each region
    pri = np.zeros(int(rho[-1]), np.int32)
    pri[rho[:-1] + ref] = 1
    got = realign.best_alleles_breaking_ties(hip_engine, _B(rro, rho, oo), lk, pri)
    for g in range(len(nr)):
        m = lk[int(oo[g]):int(oo[g + 1])].reshape(nr[g], nh[g])
        for r in range(nr[g]):
            b = int(np.argmax(m[r]))
            override = ref[g] != b and m[r, b] - m[r, ref[g]] < 0.2
            assert got.allele_index[rro[g] + r] == (ref[g] if override else b)


def test_indexed_alignment_equals_the_plain_call(hip_engine):
    rng = np.random.default_rng(4)
    alpha = b"ACGT"
    refs = [bytes(alpha[int(x)] for x in rng.integers(0, 4, int(rng.integers(80, 400)))) for _ in range(7)]
    idx = rng.integers(0, len(refs), 300)
    alts = []
    for a in idx:
        s = int(rng.integers(0, len(refs[a]) - 40))
        read = bytearray(refs[a][s:s + int(rng.integers(30, 150))])
        for q in rng.integers(0, len(read), 3):
            read[q] = alpha[int(rng.integers(0, 4))]
        alts.append(bytes(read))
    al = SmithWatermanAligner(hip_engine)
    plain = al.align_batch([(refs[a], alt) for a, alt in zip(idx, alts)], STANDARD_NGS, "SoftClip")
    idx2 = idx.copy()
    idx2[::11] = -1
    try:
        for chunks in (0, 3):
            hip_engine.set_switch("sw_chunks", chunks)
            got = al.align_indexed(refs, alts, idx2, STANDARD_NGS, "SoftClip")
            for a, (g, p) in enumerate(zip(got, plain)):
                assert (g is None) if idx2[a] < 0 else g == p, a
    finally:
        hip_engine.set_switch("sw_chunks", 0)
    with pytest.raises(PhmmError, match="out of range"):
        al.align_indexed(refs, alts[:3], [0, 7, 1], STANDARD_NGS, "SoftClip")


@pytest.mark.parametrize("with_keep", [False, True])
def test_realign_to_best_equals_the_two_steps(hip_engine, with_keep):
    """Likelihoods of a synthetic batch (mixed read lengths, 1 ... 8 haplotypes) -> best allele with the reference's
    haplotype priorities -> SoftClip alignment of every read to that haplotype: equal to oracle_best_alleles followed by
    the oracle's Smith-Waterman, read by read."""
    parts = [synthetic.make_regions(1, int(nr), int(nh), 260, [60, 100, 150], seed=50 + k)
             for k, (nr, nh) in enumerate([(24, 1), (40, 3), (31, 8), (17, 5), (3, 2)])]
    b = RegionBatch.concat(parts)
    lk = hip_engine.compute(b)
    rng = np.random.default_rng(1)
    pri = realign.haplotype_alignment_tiebreaking_priority(np.arange(b.n_haps) % 3 == 0, rng.integers(1, 4, b.n_haps))
    keep = (rng.random(b.n_reads) > 0.15).astype(np.uint8) if with_keep else None
    prm = ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS
    best, res = realign.realign_reads_to_their_best_haplotype(hip_engine, b, lk, pri, keep)
    nr = np.diff(b.region_read_off.astype(np.int64))
    nh = np.diff(b.region_hap_off.astype(np.int64))
    wb, wl, wc = _oracle_best(nr, nh, b.region_read_off, b.region_hap_off, b.out_off, lk, pri, keep, 0.2)
    assert np.array_equal(best.allele_index, wb) and np.array_equal(best.likelihood, wl)
    assert np.array_equal(best.confidence, wc, equal_nan=True)
    reg = np.repeat(np.arange(b.n_regions), nr)
    n_aligned = 0
    for r in range(b.n_reads):
        if wb[r] < 0:
            assert res[r] is None
            continue
        hp = int(b.region_hap_off[reg[r]]) + int(wb[r])
        hap = b.hap_bases[int(b.hap_off[hp]):int(b.hap_off[hp + 1])]
        read = b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])]
        cig, off = oracle.sw_align(hap, read, [prm.match_value, prm.mismatch_penalty, prm.gap_open_penalty, prm.gap_extend_penalty], "SoftClip")
        assert res[r].alignment_offset == off and np.array_equal(res[r].elements, cig), r
        n_aligned += 1
    assert n_aligned > 80
    # and the same through the separate calls
    sep = realign.best_alleles_breaking_ties(hip_engine, b, lk, pri, keep)
    assert np.array_equal(sep.allele_index, best.allele_index)


def test_realign_argument_errors(hip_engine):
    b = synthetic.make_regions(2, 4, 2, 80, 40, seed=3)
    lk = hip_engine.compute(b)
    with pytest.raises(PhmmError, match="too little room"):
        bad = RegionBatch(**{**{f: getattr(b, f) for f in RegionBatch.FIELDS}, "out_off": (b.out_off // 2).astype(np.uint64)})
        realign.best_alleles_breaking_ties(hip_engine, bad, lk)
    with pytest.raises(PhmmError, match="threshold"):
        realign.best_alleles_breaking_ties(hip_engine, b, lk, threshold=float("nan"))
    best, res = realign.realign_reads_to_their_best_haplotype(hip_engine, b, lk, parameters=NEW_SW_PARAMETERS, capacity=1)
    assert all(r is not None for r in res)   # capacity 1 is retried with the sizes the library reports


def test_edge_cases_of_the_realign_calls(hip_engine):
    """Nothing to do, regions without alleles, every read removed: defined results, no device work where there is none."""
    import ctypes as C
    i32p = C.POINTER(C.c_int32)
    lib, h = hip_engine.lib, hip_engine._h
    assert lib.phmm_best_alleles(h, 0, None, None, None, None, None, None, 0.2, None, None, None) == _lib.PHMM_OK
    prm = ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.as_struct()
    assert lib.phmm_realign_to_best(h, 0, None, None, None, None, None, None, None, None, None, None, 0.2, C.byref(prm), 0,
                                    None, None, None, None, None, None, None) == _lib.PHMM_OK
    # three reads, no haplotype anywhere: search_best_allele's None (allele_likelihoods.rs:465-475), nothing aligned
    rro, rho = np.array([0, 3], np.uint32), np.array([0, 0], np.uint32)
    read_off, reads = np.array([0, 4, 8, 12], np.uint32), np.frombuffer(b"ACGTACGTACGT", np.uint8)
    oo, cig_off = np.array([0, 0], np.uint64), np.array([0, 4, 8, 12], np.uint64)
    cigar, n_cig, off = np.zeros(12, np.uint32), np.full(3, 9, np.uint32), np.full(3, 9, np.int32)
    best, lk, conf = np.zeros(3, np.int32), np.zeros(3), np.zeros(3)
    p = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
    code = lib.phmm_realign_to_best(h, 1, p(rro, _lib.u32p), p(rho, _lib.u32p), p(read_off, _lib.u32p), p(reads, _lib.u8p),
                                    p(np.zeros(1, np.uint32), _lib.u32p), None, p(oo, _lib.u64p), None, None, None, 0.2, C.byref(prm), 0,
                                    p(cig_off, _lib.u64p), p(cigar, _lib.u32p), p(n_cig, _lib.u32p), p(off, i32p), p(best, i32p),
                                    p(lk, _lib.f64p), p(conf, _lib.f64p))
    assert code == _lib.PHMM_OK, hip_engine.last_error()
    assert best.tolist() == [-1, -1, -1] and np.all(np.isneginf(lk)) and np.all(np.isnan(conf)) and n_cig.tolist() == [0, 0, 0]
    # every read removed by the filter: no best allele, no alignment
    b = synthetic.make_regions(2, 5, 3, 80, 40, seed=4)
    got, res = realign.realign_reads_to_their_best_haplotype(hip_engine, b, hip_engine.compute(b), keep=np.zeros(b.n_reads, np.uint8))
    assert np.all(got.allele_index == -1) and all(r is None for r in res)
