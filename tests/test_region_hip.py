"""phmm_region_compute / phmm_region_submit on the MI355X: the whole per-region path in one enqueue -- pre-step, PairHMM,
normalisation + disqualification, best alleles, Smith-Waterman to the best haplotype, projection onto the reference --
EQUAL, field by field, to phmm_engine_compute followed by phmm_realign_reads (the two calls it fuses), and to the oracle
pipeline built from the pieces the reference's own tests pin (tests/test_engine_oracle.py, test_best_alleles_oracle.py,
test_sw_oracle.py, test_cigar_oracle.py).  Reference: src/haplotype/haplotype_caller_engine.rs:1311-1357."""
import ctypes as C
import threading

import numpy as np
import pytest

from lorikeet_amd import _lib, realign, region, synthetic
from lorikeet_amd.batch import Read, RegionBatch
from lorikeet_amd.engine import HipPairHMMEngine, PhmmError
from oracle import oracle

from project_scenarios import oracle_read as _oracle_read, scenario as _scenario

pytestmark = pytest.mark.gpu
MODELS = ["none", "hostile", "aggressive", "conservative"]


def _cfg(pcr=3, symmetric=True, dynamic=False, gcp=10, cap=-4.5, disable_cap=False, bq=18, scale=1.0, err=0.02):
    c = _lib.EngineConfig()
    c.constant_gcp, c.pcr_error_model, c.base_quality_score_threshold = gcp, pcr, bq
    c.dynamic_read_disqualification, c.symmetrically_normalize_alleles_to_reference = int(dynamic), int(symmetric)
    c.disable_cap_read_qualities_to_mapq = int(disable_cap)
    c.log10_global_read_mismapping_rate, c.read_disqualification_scale, c.expected_error_rate_per_base = cap, scale, err
    return c


def _engine_compute(eng, cfg, b, mapq, ref_hap):
    p = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
    out, keep = np.full(b.n_out, np.nan), np.zeros(b.n_reads, np.uint8)
    mq, rr = np.ascontiguousarray(mapq, np.uint8), np.ascontiguousarray(ref_hap, np.int32)
    code = eng.lib.phmm_engine_compute(eng._h, C.byref(cfg), b.n_regions, p(b.region_read_off, _lib.u32p), p(b.region_hap_off, _lib.u32p),
                                       p(b.read_off, _lib.u32p), p(b.read_bases, _lib.u8p), p(b.base_q, _lib.u8p), p(b.ins_q, _lib.u8p),
                                       p(b.del_q, _lib.u8p), p(mq, _lib.u8p), p(b.hap_off, _lib.u32p), p(b.hap_bases, _lib.u8p),
                                       p(rr, C.POINTER(C.c_int32)), p(b.out_off, _lib.u64p), p(out, _lib.f64p), p(keep, _lib.u8p))
    if code != _lib.PHMM_OK:
        raise PhmmError(code, eng.last_error())
    return out, keep


def _same(got, out, keep, best, proj, exact=True):
    if exact:
        assert np.array_equal(got.likelihoods, out)
    else:
        assert np.max(np.abs(got.likelihoods - out)) < 1e-11
    assert np.array_equal(got.keep, keep.astype(bool))
    assert np.array_equal(got.best.allele_index, best.allele_index)
    if exact:
        assert np.array_equal(got.best.likelihood, best.likelihood) and np.array_equal(got.best.confidence, best.confidence, equal_nan=True)
    assert np.array_equal(got.reads.status, proj.status) and np.array_equal(got.reads.new_pos, proj.new_pos)
    for r in range(len(proj.cigars)):
        assert np.array_equal(got.reads.cigars[r], proj.cigars[r]), r


def _priorities(b, hap_cigars, ref_hap):
    is_ref = np.zeros(b.n_haps, np.int32)
    for g in range(b.n_regions):
        is_ref[int(b.region_hap_off[g]) + ref_hap[g]] = 1
    return realign.haplotype_alignment_tiebreaking_priority(is_ref, [len(c) for c in hap_cigars])


def _noisy_quals(b, seed):
    """Scenario reads come with flat Q30 / Q45: give them the spread the pre-step reacts to."""
    rng = np.random.default_rng(seed)
    b.base_q[:] = rng.choice([37, 32, 27, 22, 12, 6, 2], len(b.base_q), p=[.5, .15, .1, .1, .08, .05, .02])
    b.ins_q[:] = rng.integers(20, 46, len(b.ins_q))
    b.del_q[:] = rng.integers(20, 46, len(b.del_q))
    return rng.choice([60, 60, 40, 25, 10], b.n_reads).astype(np.uint8)


@pytest.mark.parametrize("seed,pcr,symmetric,dynamic,low", [(1, 3, True, False, False), (2, 0, False, True, False), (3, 1, True, True, True),
                                                             (4, 2, False, False, True), (5, 3, True, False, False)])
def test_region_call_equals_the_two_calls_it_fuses(hip_engine, seed, pcr, symmetric, dynamic, low):
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(seed, n_regions=7, low_complexity=low)
    mapq = _noisy_quals(b, seed)
    cfg = _cfg(pcr=pcr, symmetric=symmetric, dynamic=dynamic)
    pri = _priorities(b, hap_cigars, ref_hap)
    out, keep = _engine_compute(hip_engine, cfg, b, mapq, ref_hap)
    best, proj = realign.realign_reads(hip_engine, b, out, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, hap_priority=pri, keep=keep)
    got = region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, hap_priority=pri)
    _same(got, out, keep, best, proj)
    assert (got.reads.status == 0).sum() > b.n_reads // 3
    # the same through the shared queue
    got2 = region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, hap_priority=pri, shared=True)
    _same(got2, out, keep, best, proj)


def _oracle_pipeline(cfg, b, mapq, ref_hap, pri):
    """engine.rs:195-242 + allele_likelihoods.rs:457-554 with the oracle's pieces -> (normalised [read][hap] flat, keep, best)."""
    out, keep, best = np.zeros(b.n_out), np.zeros(b.n_reads, bool), np.full(b.n_reads, -1, np.int32)
    for g in range(b.n_regions):
        r0, r1, h0, h1 = (int(x) for x in (b.region_read_off[g], b.region_read_off[g + 1], b.region_hap_off[g], b.region_hap_off[g + 1]))
        reads, thr = [], []
        for r in range(r0, r1):
            s, e = int(b.read_off[r]), int(b.read_off[r + 1])
            q, i, d = oracle.modify_read_qualities(MODELS[cfg.pcr_error_model], b.read_bases[s:e], int(mapq[r]), b.base_q[s:e], b.ins_q[s:e], b.del_q[s:e],
                                                   cfg.base_quality_score_threshold, bool(cfg.disable_cap_read_qualities_to_mapq))
            reads.append(Read(b.read_bases[s:e], q, i, d, np.full(e - s, cfg.constant_gcp, np.uint8)))
            thr.append(oracle.read_disqualification_threshold(b.base_q[s:e], bool(cfg.dynamic_read_disqualification), cfg.read_disqualification_scale,
                                                              cfg.expected_error_rate_per_base))
        haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(h0, h1)]
        rb = RegionBatch.from_regions([(reads, haps)])
        raw = oracle.compute_batch(rb.as_dict(), n_threads=4).reshape(r1 - r0, h1 - h0)
        norm = oracle.normalize_likelihoods(raw.T.copy(), cfg.log10_global_read_mismapping_rate, bool(cfg.symmetrically_normalize_alleles_to_reference), ref_hap[g])
        _, kp, _ = oracle.filter_poorly_modeled_evidence(norm.copy(), thr)
        out[int(b.out_off[g]):int(b.out_off[g]) + norm.size] = norm.T.reshape(-1)
        keep[r0:r1] = kp
        wb, _, _ = oracle.best_alleles(norm, pri[h0:h1], 0.2)
        best[r0:r1] = np.where(kp, wb, -1)
    return out, keep, best


@pytest.mark.parametrize("seed,low", [(21, False), (22, True)])
def test_region_call_equals_the_oracle_pipeline(hip_engine, seed, low):
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(seed, n_regions=5, low_complexity=low)
    mapq = _noisy_quals(b, seed)
    cfg = _cfg(pcr=3, dynamic=True)
    pri = _priorities(b, hap_cigars, ref_hap)
    got = region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, hap_priority=pri)
    out, keep, best = _oracle_pipeline(cfg, b, mapq, ref_hap, pri)
    assert np.max(np.abs(got.likelihoods - out)) < 1e-9
    assert np.array_equal(got.keep, keep) and np.array_equal(got.best.allele_index, best)
    reg = np.repeat(np.arange(b.n_regions), np.diff(b.region_read_off.astype(np.int64)))
    n_ok = 0
    for r in range(b.n_reads):
        st, pos, cig = _oracle_read(b, r, reg[r], best[r], hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
        assert got.reads.status[r] == st, (r, got.reads.status[r], st)
        if st == 0:
            assert got.reads.new_pos[r] == pos and oracle.cigar_to_string(got.reads.cigars[r]) == cig, r
            n_ok += 1
    assert n_ok > b.n_reads // 3 and (~keep).sum() >= 0


def test_soft_clipped_reads_are_aligned_without_their_clips(hip_engine):
    """modify_soft_clipped_bases: the PairHMM sees the whole read, the aligner the read minus its soft clips
    (alignment_utils.rs:47-50) -- the same as the two separate calls on the two versions of the reads."""
    b, hap_cigars, hap_starts, ref_hap, ref_start, _ = _scenario(31, n_regions=5)
    rng = np.random.default_rng(31)
    regions_full, clips, orig = [], [], []
    for g in range(b.n_regions):
        reads = []
        for r in range(int(b.region_read_off[g]), int(b.region_read_off[g + 1])):
            core = bytes(b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])])
            cl, cr = (int(rng.integers(0, 9)) * int(rng.random() < 0.5) for _ in range(2))
            full = bytes(rng.choice(list(b"ACGT"), cl).astype(np.uint8)) + core + bytes(rng.choice(list(b"ACGT"), cr).astype(np.uint8))
            n = len(full)
            reads.append(Read(full, np.full(n, 30, np.uint8), np.full(n, 45, np.uint8), np.full(n, 45, np.uint8), np.full(n, 10, np.uint8)))
            clips.append((cl, cr))
            orig.append(oracle.parse_cigar(("%dS" % cl if cl else "") + "%dM" % len(core) + ("%dS" % cr if cr else "")))
        haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(int(b.region_hap_off[g]), int(b.region_hap_off[g + 1]))]
        regions_full.append((reads, haps))
    bf = RegionBatch.from_regions(regions_full)
    mapq = np.full(bf.n_reads, 60, np.uint8)
    cfg = _cfg(pcr=0)
    out, keep = _engine_compute(hip_engine, cfg, bf, mapq, ref_hap)                       # the likelihoods of the WHOLE reads
    best, proj = realign.realign_reads(hip_engine, b, out, hap_cigars, hap_starts, ref_hap, ref_start, orig, keep=keep)  # b: the clipped reads
    got = region.region_compute(hip_engine, cfg, bf, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig, read_soft_clip=np.array(clips, np.uint32))
    _same(got, out, keep, best, proj)
    assert sum(1 for c in clips if c != (0, 0)) > 5


def test_single_allele_regions_are_not_realigned_when_asked(hip_engine):
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(41, n_regions=4)
    # cut region 1 and 3 down to their reference haplotype
    regions, hc, hs = [], [], []
    for g in range(b.n_regions):
        reads = [Read(b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])], *(x[int(b.read_off[r]):int(b.read_off[r + 1])] for x in (b.base_q, b.ins_q, b.del_q, b.gcp)))
                 for r in range(int(b.region_read_off[g]), int(b.region_read_off[g + 1]))]
        a0, a1 = int(b.region_hap_off[g]), int(b.region_hap_off[g + 1])
        if g % 2:
            a1 = a0 + 1
        regions.append((reads, [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(a0, a1)]))
        hc += hap_cigars[a0:a1]
        hs += hap_starts[a0:a1]
    b2 = RegionBatch.from_regions(regions)
    mapq = np.full(b2.n_reads, 60, np.uint8)
    cfg = _cfg()
    full = region.region_compute(hip_engine, cfg, b2, mapq, hc, hs, ref_hap, ref_start, orig_cigars)
    skip = region.region_compute(hip_engine, cfg, b2, mapq, hc, hs, ref_hap, ref_start, orig_cigars, rcfg=region.realign_config(skip_single_allele=True))
    reg = np.repeat(np.arange(b2.n_regions), np.diff(b2.region_read_off.astype(np.int64)))
    single = (reg % 2) == 1
    assert np.array_equal(full.likelihoods, skip.likelihoods) and np.array_equal(full.best.allele_index, skip.best.allele_index)
    assert np.all(skip.reads.status[single] == _lib.PHMM_PROJECT_UNCHANGED) and np.any(full.reads.status[single] == 0)
    assert np.array_equal(full.reads.status[~single], skip.reads.status[~single])
    for r in np.flatnonzero(~single):
        assert np.array_equal(full.reads.cigars[r], skip.reads.cigars[r])


def _uniform_extras(b, H):
    hap_cigars = [oracle.parse_cigar("%dM" % H)] * b.n_haps
    orig = [oracle.parse_cigar("%dM" % (int(b.read_off[r + 1]) - int(b.read_off[r]))) for r in range(b.n_reads)]
    return hap_cigars, [0] * b.n_haps, [0] * b.n_regions, [1000 * (g + 1) for g in range(b.n_regions)], orig


def test_large_call_is_pipelined_transparently(hip_engine):
    """96 regions of 128 x 8 (1.8 MB per array): chunks through the three slots, equal to the one-shot path."""
    b = synthetic.make_regions(96, 128, 8, 300, [100, 150, 250], seed=5)
    hap_cigars, hs, ref_hap, ref_start, orig = _uniform_extras(b, 300)
    mapq = np.full(b.n_reads, 60, np.uint8)
    cfg = _cfg()
    with hip_engine.switches(no_pipeline=1):
        one = region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hs, ref_hap, ref_start, orig)
    got = region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hs, ref_hap, ref_start, orig)
    _same(got, one.likelihoods, one.keep, one.best, one.reads, exact=False)
    out, keep = _engine_compute(hip_engine, cfg, b, mapq, ref_hap)
    best, proj = realign.realign_reads(hip_engine, b, out, hap_cigars, hs, ref_hap, ref_start, orig, keep=keep)
    _same(got, out, keep, best, proj, exact=False)
    assert (got.reads.status == 0).sum() > b.n_reads * 0.7


def test_concurrent_workers_share_one_handle(hip_engine):
    """The reference's pattern: every worker calls with one region; the shared handle computes the waiting workers'
    regions together (phmm_region_submit / phmm_wait) -- results equal to a lone caller's."""
    T, per = 8, 6
    shared = HipPairHMMEngine(0)
    jobs = []
    for t in range(T):
        for k in range(per):
            b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(500 + 10 * t + k, n_regions=1 + (k % 3))
            jobs.append((t, b, _noisy_quals(b, t * 100 + k), hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars))
    cfg = _cfg(pcr=3)
    want = [region.region_compute(hip_engine, cfg, j[1], *j[2:]) for j in jobs]
    got = [None] * len(jobs)
    errs = []

    def worker(t):
        try:
            for i, j in enumerate(jobs):
                if j[0] == t:
                    got[i] = region.region_compute(shared, cfg, j[1], *j[2:], shared=True)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for g, w in zip(got, want):
        _same(g, w.likelihoods, w.keep, w.best, w.reads, exact=False)
    flushes, subs = shared.submit_stats()
    assert subs == len(jobs) and flushes <= subs
    shared.close()


def test_alignments_that_outgrow_the_library_slots_are_redone(hip_engine):
    """A read that needs more than 24 CIGAR elements against its haplotype: the call runs that batch again with larger slots."""
    rng = np.random.default_rng(7)
    hap = bytes(rng.choice(list(b"ACGT"), 700).astype(np.uint8))
    read = bytearray()
    for k in range(0, 660, 44):   # 15 segments, a 3-base deletion between each pair: 29 elements
        read += hap[k:k + 41]
    reads = [Read(bytes(read), np.full(len(read), 30, np.uint8), np.full(len(read), 45, np.uint8), np.full(len(read), 45, np.uint8), np.full(len(read), 10, np.uint8))]
    b = RegionBatch.from_regions([(reads, [hap])])
    extras = ([oracle.parse_cigar("700M")], [0], [0], [5000], [oracle.parse_cigar("%dM" % len(read))])
    cfg = _cfg(pcr=0, cap=-1000.0, dynamic=True, err=1.0)  # (a threshold that keeps a read with fourteen deletions)
    got = region.region_compute(hip_engine, cfg, b, np.array([60], np.uint8), *extras, capacity=64)
    assert got.keep[0]
    st, pos, cig = _oracle_read(b, 0, 0, 0, *extras)
    assert st == 0 and got.reads.status[0] == 0 and got.reads.new_pos[0] == pos and oracle.cigar_to_string(got.reads.cigars[0]) == cig
    assert cig.count("D") >= 13


def test_argument_errors(hip_engine):
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(61, n_regions=2)
    mapq = np.full(b.n_reads, 60, np.uint8)
    with pytest.raises(PhmmError) as e:
        region.region_compute(hip_engine, _cfg(pcr=7), b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG and "PCR" in str(e.value)
    with pytest.raises(PhmmError) as e:
        region.region_compute(hip_engine, _cfg(), b, mapq, hap_cigars, hap_starts, [-1, 0], ref_start, orig_cigars)
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG and "reference haplotype" in str(e.value)
    with pytest.raises(PhmmError) as e:
        region.region_compute(hip_engine, _cfg(), b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars,
                              read_soft_clip=np.full((b.n_reads, 2), 500, np.uint32))
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG and "soft clips" in str(e.value)
    with pytest.raises(PhmmError) as e:
        region.region_compute(hip_engine, _cfg(), b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, shared=True,
                              rcfg=region.realign_config(overhang_strategy=9))
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG
    # too small output slots: reported with the sizes, the mirror retries (capacity=1 first)
    got = region.region_compute(hip_engine, _cfg(), b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, capacity=1)
    want = region.region_compute(hip_engine, _cfg(), b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
    _same(got, want.likelihoods, want.keep, want.best, want.reads)


def test_two_pass_alignment_is_invisible_in_the_results(hip_engine):
    """The aligner of the region call sweeps with candidate tags only and sends the reads whose walk meets a gap through the
    full instance (switch `sw_lite`): the same results with it forced, forbidden and left to the handle, which stops taking
    the first pass for fifteen calls once a call has met gaps in more than three reads of ten."""
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(11, n_regions=6)
    mapq = _noisy_quals(b, 11)
    cfg = _cfg()
    pri = _priorities(b, hap_cigars, ref_hap)
    call = lambda: region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, hap_priority=pri)  # noqa: E731
    try:
        hip_engine.set_switch("region_sw_all", 0)  # (the chain: a call that aligns every pair beside the PairHMM takes the full instance)
        hip_engine.set_switch("sw_lite", 0)
        want = call()
        assert hip_engine.stat("sw_second_pass") == 0
        hip_engine.set_switch("sw_lite", 1)
        got = call()
        again = hip_engine.stat("sw_second_pass")
        assert 0 < again <= b.n_reads
        hip_engine.set_switch("sw_lite", -1)
        auto1 = call()
        auto2 = call()
        if again * 10 > b.n_reads * 3:
            assert hip_engine.stat("sw_second_pass") == 0      # the second call went straight to the full instance
        for g in (got, auto1, auto2):
            assert np.array_equal(g.likelihoods, want.likelihoods) and np.array_equal(g.keep, want.keep)
            assert np.array_equal(g.best.allele_index, want.best.allele_index)
            assert np.array_equal(g.reads.status, want.reads.status) and np.array_equal(g.reads.new_pos, want.reads.new_pos)
            assert all(np.array_equal(x, y) for x, y in zip(g.reads.cigars, want.reads.cigars))
    finally:
        hip_engine.set_switch("sw_lite", -1)
        hip_engine.set_switch("region_sw_all", -1)


def _equal_calls(g, want):
    assert np.array_equal(g.likelihoods, want.likelihoods) and np.array_equal(g.keep, want.keep)
    assert np.array_equal(g.best.allele_index, want.best.allele_index)
    assert np.array_equal(g.best.likelihood, want.best.likelihood) and np.array_equal(g.best.confidence, want.best.confidence, equal_nan=True)
    assert np.array_equal(g.reads.status, want.reads.status) and np.array_equal(g.reads.new_pos, want.reads.new_pos)
    assert all(np.array_equal(x, y) for x, y in zip(g.reads.cigars, want.reads.cigars))


@pytest.mark.parametrize("seed,low,n_regions,pcr,symmetric", [(71, False, 1, 3, True), (72, True, 3, 0, False), (73, False, 6, 2, True), (74, True, 1, 1, False)])
def test_aligning_every_pair_beside_the_pairhmm_is_invisible_in_the_results(hip_engine, seed, low, n_regions, pcr, symmetric):
    """A small call aligns every read against EVERY haplotype of its region on a stream of its own while pre-step and PairHMM
    run, and one kernel behind both normalises, finds the best allele and projects the alignment in that allele's slot
    (switch `region_sw_all`: pairs up to which a call goes that way, 0 = never).  Field by field the chain's results."""
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(seed, n_regions=n_regions, low_complexity=low)
    mapq = _noisy_quals(b, seed)
    cfg = _cfg(pcr=pcr, symmetric=symmetric)
    pri = _priorities(b, hap_cigars, ref_hap)
    call = lambda **kw: region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, hap_priority=pri, **kw)  # noqa: E731
    try:
        hip_engine.set_switch("region_sw_all", 0)
        n0 = hip_engine.stat("region_sw_all")
        want = call()
        assert hip_engine.stat("region_sw_all") == n0
        hip_engine.set_switch("region_sw_all", 1 << 20)
        got = call()
        assert hip_engine.stat("region_sw_all") == n0 + 1
        _equal_calls(got, want)
        skip = region.realign_config(skip_single_allele=True)
        _equal_calls(call(rcfg=skip), (hip_engine.set_switch("region_sw_all", 0), call(rcfg=skip))[1])
    finally:
        hip_engine.set_switch("region_sw_all", -1)
    assert (want.reads.status == 0).sum() >= 1


def test_every_pair_alignments_that_outgrow_the_slots_send_the_call_round_the_plain_way(hip_engine):
    """With every pair aligned, ANY pair may need more than the 24 elements reserved -- here the read against the haplotype
    that is not its best: the call is made again, the chain's way, with slots of the size needed."""
    rng = np.random.default_rng(9)
    hap = bytes(rng.choice(list(b"ACGT"), 700).astype(np.uint8))
    other = bytearray(hap)
    for k in range(20, 680, 44):  # fifteen 3-base deletions relative to `hap`
        del other[k - 3 * (k // 44):k - 3 * (k // 44) + 3]
    read = bytes(hap[5:650])
    reads = [Read(read, np.full(len(read), 30, np.uint8), np.full(len(read), 45, np.uint8), np.full(len(read), 45, np.uint8), np.full(len(read), 10, np.uint8))]
    b = RegionBatch.from_regions([(reads, [hap, bytes(other)])])
    extras = ([oracle.parse_cigar("700M"), oracle.parse_cigar("%dM" % len(other))], [0, 0], [0], [5000], [oracle.parse_cigar("%dM" % len(read))])
    cfg = _cfg(pcr=0)
    try:
        hip_engine.set_switch("region_sw_all", 0)
        want = region.region_compute(hip_engine, cfg, b, np.array([60], np.uint8), *extras)
        hip_engine.set_switch("region_sw_all", 1 << 20)
        n0 = hip_engine.stat("region_sw_all")
        got = region.region_compute(hip_engine, cfg, b, np.array([60], np.uint8), *extras)
        assert hip_engine.stat("region_sw_all") == n0 + 1
    finally:
        hip_engine.set_switch("region_sw_all", -1)
    _equal_calls(got, want)
    assert want.best.allele_index[0] == 0 and want.reads.status[0] == 0
    cig, _ = oracle.sw_align(bytes(other), read, [10, -15, -30, -5], "SoftClip")
    assert len(cig) > 24  # (the pair that does not fit)


def test_two_callers_with_private_handles_align_every_pair_at_the_same_time(hip_engine):
    """Two region calls in flight (the most for which a small call still takes the all-pairs way): each handle has queues and a
    counter of its own, and both use the same halves of the CUs -- results equal to a lone caller's, call after call."""
    jobs = []
    for t in range(2):
        for k in range(12):
            b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(900 + 50 * t + k, n_regions=1 + (k % 2))
            jobs.append((t, b, _noisy_quals(b, 7 * t + k), hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars))
    cfg = _cfg(pcr=2)
    hip_engine.set_switch("region_sw_all", 0)
    try:
        want = [region.region_compute(hip_engine, cfg, j[1], *j[2:]) for j in jobs]
    finally:
        hip_engine.set_switch("region_sw_all", -1)
    engines = [HipPairHMMEngine(0), HipPairHMMEngine(0)]
    for e in engines:
        e.set_switch("region_server", 0)   # (this test is about the launched pipeline's two queues; a session may hold more than five engines)
    got, errs = [None] * len(jobs), []

    def worker(t):
        try:
            for i, j in enumerate(jobs):
                if j[0] == t:
                    for _ in range(6):  # (the same call again and again: the streams and the counter keep their state)
                        got[i] = region.region_compute(engines[t], cfg, j[1], *j[2:])
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for g, w in zip(got, want):
        _equal_calls(g, w)
    assert sum(e.stat("region_sw_all") for e in engines) > len(jobs) // 2   # (most calls did go the all-pairs way)
    for e in engines:
        e.close()


def test_the_last_kernels_word_in_the_mirror_replaces_the_runtimes_wait(hip_engine):
    """A small call's thread polls a word its last kernel stores into the pinned mirror (switch `region_flag_wait`; the blocks
    of that kernel count themselves in, the last one publishes) instead of sitting in hipStreamSynchronize.  Both ways of
    waiting, the chain and the all-pairs call, many calls in a row on one handle (the counter is never reset): equal results."""
    cfg = _cfg(pcr=1)
    jobs = []
    for k in range(10):
        b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(1200 + k, n_regions=1 + (k % 3))
        jobs.append((b, _noisy_quals(b, k), hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars))
    try:
        hip_engine.set_switch("region_flag_wait", 0)
        want = [region.region_compute(hip_engine, cfg, *j) for j in jobs]
        hip_engine.set_switch("region_flag_wait", 1)
        for all_pairs in (0, 1 << 20):
            hip_engine.set_switch("region_sw_all", all_pairs)
            for _ in range(3):
                for j, w in zip(jobs, want):
                    _equal_calls(region.region_compute(hip_engine, cfg, *j), w)
    finally:
        hip_engine.set_switch("region_sw_all", -1)
        hip_engine.set_switch("region_flag_wait", -1)


def test_degenerate_regions_take_every_way_of_the_call(hip_engine):
    """Reads without haplotypes (no best allele, status 1), haplotypes without reads, one read against one haplotype, and a
    batch mixing them: the default call, the chain with the runtime's wait, and the all-pairs call with the mirror's word give
    the same fields."""
    rng = np.random.default_rng(3)
    hap = bytes(rng.choice(list(b"ACGT"), 200).astype(np.uint8))
    other = bytearray(hap)
    other[100] = ord("A") if hap[100] != ord("A") else ord("C")
    other = bytes(other)

    def rd(s, e):
        b = hap[s:e]
        return Read(b, np.full(len(b), 30, np.uint8), np.full(len(b), 45, np.uint8), np.full(len(b), 45, np.uint8), np.full(len(b), 10, np.uint8))
    whole, cig = oracle.parse_cigar("200M"), lambda n: oracle.parse_cigar("%dM" % n)  # noqa: E731
    cases = {
        "reads, no haplotypes": ([([rd(10, 110)], [])], [], [], [-1], [1000], [cig(100)]),
        "haplotypes, no reads": ([([], [hap, other])], [whole] * 2, [0, 0], [0], [1000], []),
        "one read, one haplotype": ([([rd(10, 110)], [hap])], [whole], [0], [0], [1000], [cig(100)]),
        "mixed": ([([rd(10, 110), rd(50, 180)], [hap, other]), ([], [hap]), ([rd(0, 60)], [other, hap])], [whole] * 5, [0] * 5, [0, 0, 1],
                  [1000, 2000, 3000], [cig(100), cig(130), cig(60)]),
    }
    cfg = _cfg(pcr=1)
    try:
        for name, (regs, hc, hs, ref_hap, ref_start, oc) in cases.items():
            b = RegionBatch.from_regions(regs)
            mapq = np.full(b.n_reads, 60, np.uint8)
            got = []
            for sw_all, flag in ((-1, 1), (0, 0), (1 << 20, 1)):
                hip_engine.set_switch("region_sw_all", sw_all)
                hip_engine.set_switch("region_flag_wait", flag)
                got.append(region.region_compute(hip_engine, cfg, b, mapq, hc, hs, ref_hap, ref_start, oc))
            for g in got[1:]:
                _equal_calls(g, got[0])
            if name == "reads, no haplotypes":
                assert got[0].best.allele_index.tolist() == [-1] and got[0].reads.status.tolist() == [1]
            if name == "mixed":
                assert got[0].reads.status.tolist() == [0, 0, 0] and [len(c) for c in got[0].reads.cigars] == [1, 1, 1]
    finally:
        hip_engine.set_switch("region_sw_all", -1)
        hip_engine.set_switch("region_flag_wait", -1)


def test_handles_come_and_go_with_their_hardware_queues(hip_engine):
    """Up to four handles alive on a device run their one-enqueue calls on hardware queues of their own (the device's pool:
    phmm_region.cpp), more than four go back to ordinary streams, and a destroyed handle's pair is handed to the next one:
    the same results whichever stream a call ran on, handle after handle."""
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(1400, n_regions=2)
    mapq = _noisy_quals(b, 5)
    cfg = _cfg(pcr=2)
    want = region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars)
    want_lk = hip_engine.compute(b)
    for round_ in range(3):
        engines = [HipPairHMMEngine(0) for _ in range(2 + 3 * round_)]  # 3, 6, 9 handles alive with the fixture's
        try:
            for e in engines:
                for _ in range(2):
                    _equal_calls(region.region_compute(e, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars), want)
                assert np.array_equal(e.compute(b), want_lk)
        finally:
            for e in engines:
                e.close()
    for _ in range(12):  # (the pool's indices go round)
        e = HipPairHMMEngine(0)
        try:
            _equal_calls(region.region_compute(e, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars), want)
        finally:
            e.close()


def test_a_region_of_forty_thousand_reads_equals_its_small_self_read_by_read(hip_engine):
    """SURVEY 8a trap 10: a region may hold up to --max-input-depth reads.  Every read's result depends on nothing but the read and
    its region's haplotypes, so a region made of enough copies of a small region's reads to hold 40 000 (one region: it cannot be chunked)
    must give, read by read, the small region's likelihoods, keep flags, best alleles, positions and CIGARs."""
    from lorikeet_amd.batch import RegionBatch
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(2100, n_regions=1)
    mapq = _noisy_quals(b, 21)
    cfg = _cfg(pcr=3)
    pri = _priorities(b, hap_cigars, ref_hap)
    small = region.region_compute(hip_engine, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, hap_priority=pri)
    nr, nh = b.n_reads, b.n_haps
    copies = (40000 + nr - 1) // nr
    reads = []
    for r in range(nr):
        s, e = int(b.read_off[r]), int(b.read_off[r + 1])
        reads.append(Read(b.read_bases[s:e], b.base_q[s:e], b.ins_q[s:e], b.del_q[s:e], b.gcp[s:e]))
    haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(nh)]
    deep = RegionBatch.from_regions([(reads * copies, haps)])
    assert deep.n_regions == 1 and deep.n_reads == nr * copies >= 20000
    got = region.region_compute(hip_engine, cfg, deep, np.tile(mapq, copies), hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars * copies,
                                hap_priority=pri)
    assert np.max(np.abs(got.likelihoods.reshape(copies, -1) - small.likelihoods.reshape(1, -1))) <= 1e-11   # (another lane geometry)
    assert np.array_equal(got.keep.reshape(copies, nr), np.tile(small.keep, (copies, 1)))
    assert np.array_equal(got.best.allele_index.reshape(copies, nr), np.tile(small.best.allele_index, (copies, 1)))
    assert np.array_equal(got.reads.status.reshape(copies, nr), np.tile(small.reads.status, (copies, 1)))
    assert np.array_equal(got.reads.new_pos.reshape(copies, nr), np.tile(small.reads.new_pos, (copies, 1)))
    for k in (0, 1, copies // 2, copies - 1):
        for r in range(nr):
            assert np.array_equal(got.reads.cigars[k * nr + r], small.reads.cigars[r]), (k, r)
