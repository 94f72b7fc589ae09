"""Parity tests proper: the gfx950 kernels, called through the C ABI, against the CPU oracle, the
reference's golden vectors and the reference's analytic expectations.  All need a real MI355X."""
import math
import os

import numpy as np
import pytest

import reference_cases as rc
from lorikeet_amd import HipPairHMMEngine, PhmmError, synthetic
from lorikeet_amd.batch import Read, RegionBatch
from oracle import oracle

pytestmark = pytest.mark.gpu

# f64 on both sides; differences come only from FMA contraction and the order of the last-row sum.
TOL_VS_ORACLE = 1e-9
# the reference's own gate (tests/vector_pair_hmm_unit_tests.rs:63,90)
TOL_REFERENCE = 1e-5


def _close(got, want, tol=TOL_VS_ORACLE):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    inf = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], want[inf])
    assert not np.any(np.isnan(got))
    if (~inf).any():
        assert float(np.max(np.abs(got[~inf] - want[~inf]))) <= tol


def _forced_engine(L, **kw):
    e = HipPairHMMEngine(0, **kw)
    e.set_switch("force_L", L)  # for the life of the engine
    return e


@pytest.fixture(scope="module")
def engines():
    es = {0: HipPairHMMEngine(0)}
    for L in (16, 32, 64):
        es[L] = _forced_engine(L)
    yield es
    for e in es.values():
        e.close()


@pytest.fixture(scope="module")
def engine_no_tristate():
    e = HipPairHMMEngine(0, do_not_use_tristate_correction=True)
    yield e
    e.close()


# ---------------------------------------------------------------------------------------------
# golden vectors
# ---------------------------------------------------------------------------------------------
def test_known_answer_vectors_raw_forward(kat_rows):
    """tests/vector_pair_hmm_unit_tests.rs:51-63: the raw forward() closure."""
    from lorikeet_amd.pair_hmm import forward
    for r in kat_rows:  # all 104
        got = forward(r["hap"], r["read"], r["qual"], r["ins"], r["dele"], r["gcp"])
        assert abs(got - r["expected"]) < TOL_REFERENCE


def test_known_answer_vectors_every_kernel_shape(kat_rows, engines):
    b = RegionBatch.from_regions([([Read(r["read"], r["qual"], r["ins"], r["dele"], r["gcp"])], [r["hap"]])
                                  for r in kat_rows])
    want = oracle.compute_batch(b.as_dict())
    exp = np.array([r["expected"] for r in kat_rows])
    for L, eng in engines.items():
        got = eng.compute(b)
        assert np.max(np.abs(got - exp)) < TOL_REFERENCE, L
        _close(got, want)


def test_known_answer_vectors_through_pairhmm_and_allele_likelihoods(kat_rows):
    """tests/vector_pair_hmm_unit_tests.rs:66-90: PairHMM::initialize -> compute_log10_likelihoods ->
    get_log_likelihood_array()[0], and the [allele, read] scatter."""
    from lorikeet_amd.pair_hmm import AlleleLikelihoods, Haplotype, HmmRead, PairHMM, PairHMMInputScoreImputator
    for r in kat_rows:  # all 104
        hap = Haplotype(r["hap"], True)
        read = HmmRead(r["read"], r["qual"], r["ins"], r["dele"])
        read_map = {0: [read]}
        hmm = PairHMM.initialize([hap], read_map)
        likelihoods = AlleleLikelihoods([hap], [0], read_map)
        hmm.compute_log10_likelihoods(0, likelihoods, [read], PairHMMInputScoreImputator(int(r["gcp"][0])))
        la = hmm.get_log_likelihood_array()
        assert len(la) == 1 and abs(la[0] - r["expected"]) < TOL_REFERENCE
        assert likelihoods.sample_matrix(0)[0, 0] == la[0]


# ---------------------------------------------------------------------------------------------
# the reference's analytic tests (tristate off), HIP path
# ---------------------------------------------------------------------------------------------
def _run_cases(eng, cases):
    cases = list(cases)
    b = rc.to_batch(cases)
    got = eng.compute(b)
    for c, g in zip(cases, got):
        rc.check(c, float(g))
    return b, got


def test_basic_likelihoods_hip(engine_no_tristate):
    b, got = _run_cases(engine_no_tristate, rc.basic_likelihood_cases(extensive=True))
    _close(got, oracle.compute_batch(b.as_dict(), disable_tristate=True, n_threads=8))


def test_mismatch_providers_big_reads_hip(engine_no_tristate):
    for gen in (rc.mismatch_every_position_cases(), rc.hmm_provider_cases(), rc.big_read_cases(),
                [rc.max_lengths_case()] * 3):
        b, got = _run_cases(engine_no_tristate, gen)
        _close(got, oracle.compute_batch(b.as_dict(), disable_tristate=True, n_threads=8))


def test_likelihoods_from_haplotypes_and_empty_read_list(engine_no_tristate):
    """tests/pair_hmm_unit_tests.rs:637-683."""
    from lorikeet_amd.pair_hmm import AlleleLikelihoods, Haplotype, HmmRead, PairHMM, PairHMMInputScoreImputator
    read_bases, ref_bases = b"A" * 10, b"A" * 20
    ref_h = Haplotype(ref_bases, True)
    reads = [HmmRead(read_bases, [20] * 10, [rc.MASSIVE_QUAL] * 10, [rc.MASSIVE_QUAL] * 10)]
    hmm = PairHMM.initialize([ref_h], {0: reads})
    hmm.do_not_use_tristate_correction()
    mat = AlleleLikelihoods([ref_h], [0], {0: reads})
    imp = PairHMMInputScoreImputator(rc.MASSIVE_QUAL)
    hmm.compute_log10_likelihoods(0, mat, [], imp)
    assert len(hmm.get_log_likelihood_array()) == 0
    hmm.compute_log10_likelihoods(0, mat, reads, imp)
    la = hmm.get_log_likelihood_array()
    assert len(la) == 1
    assert abs(la[0] - rc._expected_matching(10, 20, 20, rc.MASSIVE_QUAL)) <= 1e-3


def test_haplotype_indexing_inputs_equal_oracle(engine_no_tristate):
    """The caching test (pair_hmm_unit_tests.rs:725-814) pins cached == full recompute; the HIP path always
    recomputes, so it must equal the oracle's full recompute on the same haplotypes."""
    prefix, roots, reads = rc.haplotype_indexing_inputs()
    regions = []
    for read_full in reads:
        for n in range(10, len(read_full), 3):
            rd = Read(read_full[:n], [30] * n, [45] * n, [40] * n, [10] * n)
            for ps in range(len(prefix), -1, -7):
                regions.append(([rd], [prefix[ps:] + r for r in roots]))
    b = RegionBatch.from_regions(regions)
    _close(engine_no_tristate.compute(b), oracle.compute_batch(b.as_dict(), disable_tristate=True, n_threads=8))


# ---------------------------------------------------------------------------------------------
# seeded random inputs, edge cases
# ---------------------------------------------------------------------------------------------
def _random_region(rng, n_reads, n_haps, rlen, hlen, alphabet=b"ACGT", qmax=60, qmin=0):
    alpha = np.frombuffer(alphabet, np.uint8)
    haps = [alpha[rng.integers(0, len(alpha), int(rng.integers(*hlen)))] for _ in range(n_haps)]
    reads = []
    for _ in range(n_reads):
        n = int(rng.integers(*rlen))
        # ins/del quals >= 6 as the engine guarantees (...engine.rs:446-456): below Q4 the transition model is
        # improper (mi + md > 1) and results may exceed 0, which the reference asserts on (tested separately)
        reads.append(Read(alpha[rng.integers(0, len(alpha), n)], rng.integers(qmin, qmax + 1, n),
                          rng.integers(6, qmax + 1, n), rng.integers(6, qmax + 1, n), rng.integers(qmin, qmax + 1, n)))
    return reads, haps


def test_random_ragged_batches_every_kernel_shape(engines):
    rng = np.random.default_rng(1234)
    regions = []
    for _ in range(40):
        regions.append(_random_region(rng, int(rng.integers(0, 9)), int(rng.integers(1, 11)), (0, 130), (1, 420)))
    b = RegionBatch.from_regions(regions)
    want = oracle.compute_batch(b.as_dict(), n_threads=8)
    for L, eng in engines.items():
        _close(eng.compute(b), want)


def test_wildcards_raw_bytes_and_extreme_qualities(engines):
    rng = np.random.default_rng(7)
    regions = [
        _random_region(rng, 6, 5, (20, 90), (30, 200), alphabet=b"ACGTN"),        # N on both sides
        _random_region(rng, 6, 3, (20, 90), (30, 200), alphabet=b"ACGTNacgtnRY*"),  # raw byte equality, lowercase n != N
        _random_region(rng, 6, 4, (20, 90), (30, 200), qmax=255),                 # full u8 quality range
    ]
    # all-N read, all-N haplotype
    n = 25
    regions.append(([Read(b"N" * n, [30] * n, [40] * n, [40] * n, [10] * n)], [b"ACGT" * 10, b"N" * 33]))
    b = RegionBatch.from_regions(regions)
    want = oracle.compute_batch(b.as_dict(), n_threads=4)
    for L, eng in engines.items():
        _close(eng.compute(b), want)


def test_edge_shapes(engines):
    q = lambda n, v=30: np.full(n, v, np.uint8)  # noqa: E731
    e = np.zeros(0, np.uint8)
    hap304, hap305 = b"ACGT" * 76, b"ACGT" * 76 + b"A"
    regions = [
        ([Read(b"", e, e, e, e)], [b"ACGTA"]),                                  # empty read -> -inf
        ([Read(b"ACGTAACGTAAC", q(12), q(12, 40), q(12, 40), q(12, 10))], [b"ACGTA"]),  # read longer than haplotype
        ([Read(b"A", q(1), q(1, 40), q(1, 40), q(1, 10))], [b"A", b"C", b"AC"]),  # 1x1 cells
        ([], [b"ACGT"]),                                                          # empty read list: no-op
        ([Read(b"ACGT", q(4), q(4), q(4), q(4))], []),                            # no haplotypes
        ([Read(b"ACGT" * 30, q(120), q(120, 40), q(120, 45), q(120, 10))], [hap304, hap305, hap304[:303]]),  # L*K edges
    ]
    b = RegionBatch.from_regions(regions)
    want = oracle.compute_batch(b.as_dict())
    assert want[0] == -math.inf
    for L, eng in engines.items():
        _close(eng.compute(b), want)


def test_shapes_outside_the_register_kernel_use_the_generic_kernel(hip_engine):
    rng = np.random.default_rng(99)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    hap_long = alpha[rng.integers(0, 4, 2100)]  # > 64 lanes * 32 columns
    rd = lambda n: Read(alpha[rng.integers(0, 4, n)], rng.integers(6, 41, n), rng.integers(30, 46, n),  # noqa: E731
                        rng.integers(30, 46, n), np.full(n, 10))
    regions = [([rd(60), rd(45)], [hap_long, hap_long[:1500]]),
               ([rd(3000)], [alpha[rng.integers(0, 4, 150)]])]  # read too long for the LDS staging
    b = RegionBatch.from_regions(regions)
    plan = hip_engine.plan(b)
    assert plan.dominant_kernel == "phmm_forward_generic"
    plan.close()
    _close(hip_engine.compute(b), oracle.compute_batch(b.as_dict(), n_threads=4))


def test_argument_errors_mirror_the_reference_asserts(hip_engine):
    with pytest.raises(ValueError, match="same size"):  # pair_hmm.rs:425-440
        Read(b"ACGT", [30] * 4, [40] * 3, [40] * 4, [10] * 4)
    b = synthetic.make_regions(1, 4, 2, 40, 20, seed=3)
    bad = RegionBatch(**{**b.as_dict(), "out_off": np.array([0, 3], np.uint64)})  # needs 8 doubles
    with pytest.raises(PhmmError) as ei:
        hip_engine.compute(bad)
    assert ei.value.code == 1 and "out_off" in str(ei.value)
    bad = RegionBatch(**{**b.as_dict(), "read_off": b.read_off[::-1].copy()})
    with pytest.raises(PhmmError):
        hip_engine.compute(bad)


def test_positive_result_is_reported_like_the_reference_assert(hip_engine):
    """pair_hmm.rs:478-481 asserts result <= 0.  Q0 indel-open penalties make the model improper (mi = md = 1)
    and a perfectly matching read then scores log10(~2) > 0: oracle and HIP path must both refuse."""
    n = 12
    rd = Read(b"A" * n, [60] * n, [0] * n, [0] * n, [60] * n)
    b = RegionBatch.from_regions([([rd], [b"A" * 20])])
    with pytest.raises(AssertionError):
        oracle.compute_batch(b.as_dict())
    with pytest.raises(PhmmError) as ei:
        hip_engine.compute(b)
    assert ei.value.code == 4 and "cannot be greater than 0.0" in str(ei.value)
    # the engine stays usable afterwards
    ok = synthetic.make_regions(1, 2, 2, 30, 20, seed=1)
    _close(hip_engine.compute(ok), oracle.compute_batch(ok.as_dict()))


# ---------------------------------------------------------------------------------------------
# full-size shapes (BASELINE.json configs): oracle on a sample + size-independent properties
# ---------------------------------------------------------------------------------------------
def test_config2_single_region_matches_oracle(engines):
    b = synthetic.config2(1, seed=1)
    assert b.cells() == 46_080_000 and b.algorithmic_bytes() == 106_592
    want = oracle.compute_batch(b.as_dict(), n_threads=8)
    for L, eng in engines.items():
        got = eng.compute(b)
        _close(got, want)
        assert np.all(got <= 0.0)


def _permute_reads_and_haps(b, rng):
    """Same regions with reads and haplotypes shuffled inside every region; returns the batch and the
    index map from new out slots to old ones."""
    regions, remap = [], []
    for g in range(b.n_regions):
        r0, r1 = int(b.region_read_off[g]), int(b.region_read_off[g + 1])
        h0, h1 = int(b.region_hap_off[g]), int(b.region_hap_off[g + 1])
        pr, ph = rng.permutation(r1 - r0), rng.permutation(h1 - h0)
        reads = []
        for i in pr:
            s, e = int(b.read_off[r0 + i]), int(b.read_off[r0 + i + 1])
            reads.append(Read(b.read_bases[s:e], b.base_q[s:e], b.ins_q[s:e], b.del_q[s:e], b.gcp[s:e]))
        haps = [b.hap_bases[int(b.hap_off[h0 + j]):int(b.hap_off[h0 + j + 1])] for j in ph]
        regions.append((reads, haps))
        base = int(b.out_off[g])
        remap.append(base + (pr[:, None] * (h1 - h0) + ph[None, :]).reshape(-1))
    return RegionBatch.from_regions(regions), np.concatenate(remap)


def test_config3_mixed_read_lengths_properties(hip_engine):
    b = synthetic.config3(96, seed=20250928)
    got = hip_engine.compute(b)
    assert np.all(got <= 0.0) and not np.any(np.isnan(got))
    # (1) oracle on a sample of regions
    sub = b.region_slice(40, 44)
    _close(got[int(b.out_off[40]):int(b.out_off[44])], oracle.compute_batch(sub.as_dict(), n_threads=8))
    # (2) each pair is independent: shuffling reads / haplotypes inside regions permutes the output, bit for bit
    pb, remap = _permute_reads_and_haps(b, np.random.default_rng(5))
    assert np.array_equal(hip_engine.compute(pb), got[remap])
    # (3) batching is transparent: region-by-region calls (a different kernel shape: few pairs -> more lanes per
    #     pair) agree with the batched call to the last-row summation order
    for g in (0, 17, 95):
        one = hip_engine.compute(b.region_slice(g, g + 1))
        _close(one, got[int(b.out_off[g]):int(b.out_off[g + 1])], tol=1e-12)
    # (4) a read is best explained by a haplotype it was copied from: the top likelihood beats the median
    m = got.reshape(-1, 8)
    assert np.all(m.max(axis=1) >= np.median(m, axis=1))


def test_config5_stress_shape(hip_engine):
    b = synthetic.config5(2, seed=7000)  # 512 reads x 64 haplotypes, R=150, H=400
    assert b.cells() == 2 * 512 * 150 * 64 * 400
    got = hip_engine.compute(b)
    assert np.all(got <= 0.0) and not np.any(np.isnan(got))
    # oracle on a sample: 24 reads of region 1 against all 64 haplotypes, one read per oracle task
    g = 1
    h0 = int(b.region_hap_off[g])
    haps = [b.hap_bases[int(b.hap_off[h0 + j]):int(b.hap_off[h0 + j + 1])] for j in range(64)]
    regions, rows = [], []
    for i in range(0, 512, 22):
        r = int(b.region_read_off[g]) + i
        s, e = int(b.read_off[r]), int(b.read_off[r + 1])
        regions.append(([Read(b.read_bases[s:e], b.base_q[s:e], b.ins_q[s:e], b.del_q[s:e], b.gcp[s:e])], haps))
        rows.append(got[int(b.out_off[g]) + i * 64:int(b.out_off[g]) + (i + 1) * 64])
    want = oracle.compute_batch(RegionBatch.from_regions(regions).as_dict(), n_threads=8)
    _close(np.concatenate(rows), want)


def test_host_path_size_classes(hip_engine):
    """phmm_compute stages small batches in one shot (the smallest without a D2H copy: the kernels write the pinned
    mirror), and cuts everything above 0.5 MB per array into pipelined chunks of growing size (0.5 MB x 4, 1, 1, 2, 2,
    4 ... MB) whose results are fetched after the kernels are seen to finish: every size class, and the one-shot path
    forced on a large batch, against the oracle and against each other."""
    b = synthetic.config2(600, seed=41)      # 11.5 MB per per-base array: ten chunks, all sizes up to 4 MB
    chunked = hip_engine.compute(b)
    with hip_engine.switches(no_pipeline=1):
        one_shot = hip_engine.compute(b)     # the same batch in one shot
    _close(one_shot, chunked, tol=1e-12)     # chunks plan their own kernel shapes: last-row summation order only
    _close(chunked[:int(b.out_off[3])], oracle.compute_batch(b.region_slice(0, 3).as_dict(), n_threads=8))
    # 1 region (results written by the kernels into the mirror), 8 (64 KB of results: the last such size), 9 (first with
    # a D2H copy), 26 / 28 (either side of the one-shot limit), 40 (two chunks), 300 (eight chunks)
    for n in (1, 8, 9, 26, 28, 40, 300):
        sub = b.region_slice(100, 100 + n)
        got = hip_engine.compute(sub)
        _close(got, chunked[int(b.out_off[100]):int(b.out_off[100 + n])], tol=1e-12)


def test_split_phase_api_on_a_torch_stream(hip_engine):
    import torch
    b = synthetic.config2(8, seed=3)
    want = hip_engine.compute(b)
    plan = hip_engine.plan(b)
    assert plan.cells == b.cells() and plan.algorithmic_bytes == b.algorithmic_bytes()
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(getattr(b, k)).to(dev) for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    out = torch.zeros(b.n_out, dtype=torch.float64, device=dev)
    plan.bind_torch(t, out)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.launch(st.cuda_stream)
    st.synchronize()
    plan.status()
    assert np.array_equal(out.cpu().numpy(), want)
    # batch-owned buffers
    plan2 = hip_engine.plan(b)
    plan2.upload()
    plan2.launch()
    assert np.array_equal(plan2.download(), want)
    plan.close()
    plan2.close()


def test_concurrent_host_threads_one_handle_each():
    """The reference calls the path from up to --threads rayon workers, each with its own engine clone
    (assembly_region_walker.rs:227).  Same here: one phmm_handle per thread, calls overlap (ctypes drops the
    GIL), every thread must get exactly the single-threaded answer."""
    import threading
    n_threads, n_iter = 8, 12
    batches = [synthetic.make_regions(3, 24, 1 + (t % 5), 120 + 13 * t, [40, 77, 101], seed=500 + t) for t in range(n_threads)]
    ref_eng = HipPairHMMEngine(0)
    want = [ref_eng.compute(b) for b in batches]
    ref_eng.close()
    errors = []

    def worker(t):
        try:
            eng = HipPairHMMEngine(0)
            for i in range(n_iter):
                got = eng.compute(batches[(t + i) % n_threads])
                if not np.array_equal(got, want[(t + i) % n_threads]):
                    errors.append((t, i))
            eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


# ---------------------------------------------------------------------------------------------
# chained kernel (phmm_forward_chain<L,K>): reads of a region stream back to back through the lanes
# ---------------------------------------------------------------------------------------------
@pytest.fixture()
def force_chain(engines):
    for e in engines.values():
        e.set_switch("force_chain", 5)  # runs of 5 reads per wave regardless of batch size
    yield
    for e in engines.values():
        e.set_switch("force_chain", -1)


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_chained_kernel_matches_oracle(engines, force_chain, kat_rows, lanes):
    hip_engine = engines[lanes]  # forced lanes per pair (the planner would give these small batches 64)
    rng = np.random.default_rng(77)
    regions = [_random_region(rng, int(rng.integers(1, 14)), int(rng.integers(1, 10)), (1, 140), (1, 300)) for _ in range(30)]
    b = RegionBatch.from_regions(regions)
    plan = hip_engine.plan(b)
    assert plan.dominant_kernel.replace("chain_k<", "chain<").startswith("phmm_forward_chain<%d," % lanes)
    plan.close()
    _close(hip_engine.compute(b), oracle.compute_batch(b.as_dict(), n_threads=8))
    # the reference's known-answer vectors, all in ONE region per haplotype so that reads really chain
    by_hap = {}
    for r in kat_rows:
        by_hap.setdefault(r["hap"], []).append(r)
    regs = [([Read(r["read"], r["qual"], r["ins"], r["dele"], r["gcp"]) for r in rows], [hap]) for hap, rows in by_hap.items()]
    kb = RegionBatch.from_regions(regs)
    got = hip_engine.compute(kb)
    exp = np.array([r["expected"] for rows in by_hap.values() for r in rows])
    assert np.max(np.abs(got - exp)) < TOL_REFERENCE
    # mixed read lengths at full shape, sample of regions against the oracle
    c3 = synthetic.config3(6, seed=11)
    _close(hip_engine.compute(c3), oracle.compute_batch(c3.as_dict(), n_threads=8))


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_chained_kernel_falls_back_exactly(engines, force_chain, lanes):
    hip_engine = engines[lanes]
    """Haplotypes with 'N' and reads with gcp == 0 cannot use the chained fast path: same wave, plain sweep."""
    rng = np.random.default_rng(78)
    regions = [_random_region(rng, 7, 4, (5, 90), (30, 200), alphabet=b"ACGTN"),  # N in haplotypes (and reads)
               _random_region(rng, 7, 3, (5, 90), (30, 200))]
    for rd in regions[1][0][::2]:
        rd.gcp[len(rd.gcp) // 2] = 0  # im = 0 on one row
    b = RegionBatch.from_regions(regions)
    plan = hip_engine.plan(b)
    assert plan.dominant_kernel.replace("chain_k<", "chain<").startswith("phmm_forward_chain<")
    plan.close()
    _close(hip_engine.compute(b), oracle.compute_batch(b.as_dict(), n_threads=4))


@pytest.mark.parametrize("lanes", [16, 32])
def test_chained_kernel_long_reads(engines, force_chain, lanes):
    """Reads longer than the 256-row LDS ring stream through it; with an 'N' haplotype, a gcp == 0 or a base
    quality 0 in the chain, the in-wave general path builds the rows of such reads on the fly."""
    hip_engine = engines[lanes]
    rng = np.random.default_rng(79)
    regions = [_random_region(rng, 6, 3, (230, 700), (40, 250), qmin=1),                      # streamed
               _random_region(rng, 5, 2, (230, 700), (40, 250), alphabet=b"ACGTN", qmin=1),   # general path, unstaged
               _random_region(rng, 5, 3, (250, 600), (40, 250), qmin=1)]
    regions[2][0][1].gcp[3] = 0
    regions[2][0][3].quals[100] = 0  # match prior 0: the row form with the prior folded out cannot be used
    b = RegionBatch.from_regions(regions)
    plan = hip_engine.plan(b)
    assert plan.dominant_kernel.replace("chain_k<", "chain<").startswith("phmm_forward_chain<")
    plan.close()
    _close(hip_engine.compute(b), oracle.compute_batch(b.as_dict(), n_threads=8))


@pytest.mark.parametrize("streams", [2, 4])
def test_chained_kernel_streams(engines, force_chain, streams, kat_rows):
    """16 lanes per pair: the run of reads is split into 2 or 4 streams swept side by side on 2 or 1 haplotype slots
    each (what the planner does for regions whose haplotype count is not a multiple of four)."""
    hip_engine = engines[16]
    hip_engine.set_switch("force_streams", streams)
    try:
        rng = np.random.default_rng(80 + streams)
        # 1..9 haplotypes, 1..14 reads (fewer reads than streams, uneven sub-runs), reads beyond the per-stream ring
        regions = [_random_region(rng, int(rng.integers(1, 15)), int(rng.integers(1, 10)), (1, 140), (1, 300), qmin=1)
                   for _ in range(30)]
        regions.append(_random_region(rng, 7, 3, (60, 400), (40, 250), qmin=1))
        b = RegionBatch.from_regions(regions)
        plan = hip_engine.plan(b)
        assert plan.dominant_kernel.endswith("x%d streams" % streams), plan.dominant_kernel
        plan.close()
        _close(hip_engine.compute(b), oracle.compute_batch(b.as_dict(), n_threads=8))
        # the general path inside a split run: 'N' haplotypes, gcp == 0, base quality 0
        regions = [_random_region(rng, 9, 3, (5, 90), (30, 200), alphabet=b"ACGTN"), _random_region(rng, 9, 2, (5, 90), (30, 200))]
        b = RegionBatch.from_regions(regions)
        _close(hip_engine.compute(b), oracle.compute_batch(b.as_dict(), n_threads=4))
        # the reference's known-answer vectors, one region per haplotype (a single haplotype: 4 streams fill the wave)
        by_hap = {}
        for r in kat_rows:
            by_hap.setdefault(r["hap"], []).append(r)
        regs = [([Read(r["read"], r["qual"], r["ins"], r["dele"], r["gcp"]) for r in rows], [hap]) for hap, rows in by_hap.items()]
        got = hip_engine.compute(RegionBatch.from_regions(regs))
        exp = np.array([r["expected"] for rows in by_hap.values() for r in rows])
        assert np.max(np.abs(got - exp)) < TOL_REFERENCE
    finally:
        hip_engine.set_switch("force_streams", 0)


@pytest.mark.parametrize("nh", [1, 2])
def test_streams_of_uneven_runs(engines, nh):
    """Runs of up to twelve reads of very different lengths swept as four / two streams side by side (one / two haplotypes): one
    long read first, last, in the middle; a run shorter than its streams; equal reads -- against the oracle, read by read.  (Round
    6 built cuts of a run by ROWS instead of by count of reads with these cases and took them out again: fewer swept cells, slower
    small launches and host-buffer calls -- NOTEBOOK.md 20.6.)"""
    eng = engines[16]
    eng.set_switch("force_chain", 12)
    try:
        rng = np.random.default_rng(600 + nh)
        alpha = np.frombuffer(b"ACGT", np.uint8)

        def region(lens):
            haps = [alpha[rng.integers(0, 4, int(rng.integers(100, 300)))] for _ in range(nh)]
            return [Read(alpha[rng.integers(0, 4, n)], rng.integers(2, 60, n), rng.integers(6, 60, n), rng.integers(6, 60, n), rng.integers(2, 60, n))
                    for n in lens], haps

        shapes = [[250] + [30] * 11, [30] * 11 + [250], [30] * 5 + [250] + [30] * 6, [250, 250, 30, 30, 30], [100] * 12, [77], [200, 10],
                  [1, 1, 1, 300, 1, 1, 1], list(range(10, 130, 10)), [300] * 12]
        regions = [region(lens) for lens in shapes for _ in range(3)]
        b = RegionBatch.from_regions(regions)
        plan = eng.plan(b)
        assert plan.dominant_kernel.endswith("x%d streams" % (4 if nh == 1 else 2)), plan.dominant_kernel
        plan.close()
        _close(eng.compute(b), oracle.compute_batch(b.as_dict(), n_threads=8))
    finally:
        eng.set_switch("force_chain", -1)


def test_planner_fills_the_wave_for_any_haplotype_count(hip_engine):
    """Large batches of regions with 1, 2, 3, 5 and 6 haplotypes: chained with 4, 2, 4, 4 and 2 streams."""
    for nh, streams in ((1, 4), (2, 2), (3, 4), (5, 4), (6, 2), (8, 1)):
        b = synthetic.make_regions(2400, 64, nh, 120, 60, seed=100 + nh)  # enough wave-sweeps for the batch to chain
        plan = hip_engine.plan(b)
        want = "x%d streams" % streams if streams > 1 else ">"
        assert plan.dominant_kernel.replace("chain_k<", "chain<").startswith("phmm_forward_chain<16,") and plan.dominant_kernel.endswith(want), (nh, plan.dominant_kernel)
        plan.close()
        sub = b.region_slice(0, 6)
        got = hip_engine.compute(b)
        _close(got[:int(b.out_off[6])], oracle.compute_batch(sub.as_dict(), n_threads=8))


def test_chained_and_plain_kernels_agree(engines):
    hip_engine = engines[16]
    b = synthetic.config2(24, seed=9)
    plain = hip_engine.compute(b)
    with hip_engine.switches(force_chain=16):
        chained = hip_engine.compute(b)
    _close(chained, plain, tol=1e-12)
    with hip_engine.switches(force_chain=0):
        assert np.array_equal(hip_engine.compute(b), plain)
