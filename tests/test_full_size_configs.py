"""BASELINE.json's full-size sets, every pair of them (configs 3 and 5 were parity-tested on a few regions only).

The scalar oracle needs minutes per set, so the full sets are cross-checked against the SIMD stand-in of the reference's
vector arm (oracle/pairhmm_simd.c: an independent implementation -- f32 first, f64 redo, inter-pair lanes -- pinned by the
reference's 104 vectors at 1e-5, tests/test_oracle_simd.py) at the reference's own gate for that arm, 1e-5 on EVERY pair,
and against the scalar oracle at 1e-9 on a sample of regions spread over the set; plus size-independent properties:
the resident launch and the chunked host path agree, a permutation of the regions permutes the results."""
import numpy as np
import pytest

from lorikeet_amd import synthetic
from oracle import oracle

pytestmark = pytest.mark.gpu


def _check_full_set(eng, b, sample):
    got = eng.compute(b)                                    # host buffers: chunked pipeline, a plan per chunk
    plan = eng.plan(b)                                      # resident: one plan over the whole set
    plan.upload()
    plan.launch()
    resident = plan.download()
    plan.close()
    assert got.shape == (b.n_out,) and (got <= 0).all() and np.isfinite(got).all()
    assert np.max(np.abs(got - resident)) <= 1e-12          # chunks choose run lengths of their own: summation order only
    simd, redone = oracle.compute_batch_simd(b.as_dict(), n_threads=16, native=False)
    assert float(np.max(np.abs(got - simd))) <= 1e-5, "full set vs the vector-arm stand-in"
    for g in sample:
        sub = b.region_slice(g, g + 1)
        want = oracle.compute_batch(sub.as_dict(), n_threads=16)
        have = got[int(b.out_off[g]):int(b.out_off[g + 1])]
        assert float(np.max(np.abs(have - want))) <= 1e-9, g
    return got


def test_config3_all_10000_regions(hip_engine):
    b = synthetic.config3()
    assert b.n_regions == 10000 and abs(b.cells() - 5.12e11) < 0.01e11
    got = _check_full_set(hip_engine, b, [0, 1, 4999, 7777, 9999])
    # regions 2000..2999 moved to the front: the same numbers, moved
    from lorikeet_amd.batch import RegionBatch
    perm = RegionBatch.concat([b.region_slice(2000, 3000), b.region_slice(0, 2000)])
    p = hip_engine.compute(perm)
    o = b.out_off.astype(np.int64)
    assert np.max(np.abs(p[:o[3000] - o[2000]] - got[o[2000]:o[3000]])) <= 1e-12
    assert np.max(np.abs(p[o[3000] - o[2000]:] - got[:o[2000]])) <= 1e-12


def test_config5_all_256_stress_regions(hip_engine):
    b = synthetic.config5()
    assert b.n_regions == 256 and b.cells() == 256 * 512 * 150 * 64 * 400    # 5.03e11, SURVEY 8(d)
    _check_full_set(hip_engine, b, [0, 255])


def test_ragged_mix_all_1536_regions(hip_engine):
    """The bench's long-tailed mix (3 ... 5 000 reads, 1 ... 128 haplotypes of 60 ... 500 bases, reads of 30 ... 250 mixed
    inside a region, 'N' runs): every one of its 1.57 M pairs, through every kernel class the planner picks for it -- per-read
    kernels, the chained kernels of all four K ranges at 16 and 32 lanes per pair, multi-stream items for haplotype
    remainders -- and the scalar oracle on the smallest, the largest and a few regions in between."""
    b = synthetic.ragged()
    assert b.n_regions == 1536
    cells = np.diff(b.region_read_off.astype(np.int64)) * np.diff(b.region_hap_off.astype(np.int64))
    order = np.argsort(cells)
    sample = [int(order[0]), int(order[len(order) // 4]), int(order[len(order) // 2]), int(order[-40]), int(order[-1])]
    _check_full_set(hip_engine, b, sample)


def test_one_region_at_the_reference_maximum_depth(hip_engine):
    """SURVEY 8a trap 10: a region may hold up to --max-input-depth reads (200 000, cli.rs) against up to 128 haplotypes
    (cli.rs:1588-1591).  One region of 70 000 short reads x 128 haplotypes (9 M pairs: past every 16-bit count) and one of 200 000 x 3,
    alone in a call -- a single region cannot be cut into chunks -- through the host path, the resident path and, for the engine-
    level call's pre- and post-step, phmm_engine_compute: the SIMD stand-in at 1e-5 on every pair, the scalar oracle at 1e-9 on
    the first and last reads, every read's row a permutation-free copy of the same read in a small region."""
    from lorikeet_amd.batch import RegionBatch
    for nr, nh, H, R, seed in ((70000, 128, 64, [20, 30, 40], 31), (200000, 3, 120, [30, 50], 32)):
        b = synthetic.make_regions(1, nr, nh, H, R, seed=seed)
        assert b.n_regions == 1 and b.n_reads == nr and b.n_out == nr * nh
        got = hip_engine.compute(b)
        assert got.shape == (nr * nh,) and (got <= 0).all() and np.isfinite(got).all()
        plan = hip_engine.plan(b)
        plan.upload()
        plan.launch()
        assert np.max(np.abs(plan.download() - got)) <= 1e-12
        plan.close()
        simd, _ = oracle.compute_batch_simd(b.as_dict(), n_threads=16, native=False)
        assert float(np.max(np.abs(got - simd))) <= 1e-5
        # the first and the last 40 reads as a region of their own, scalar oracle
        from lorikeet_amd.batch import Read
        for r0 in (0, nr - 40):
            reads = []
            for r in range(r0, r0 + 40):
                s0, s1 = int(b.read_off[r]), int(b.read_off[r + 1])
                reads.append(Read(b.read_bases[s0:s1], b.base_q[s0:s1], b.ins_q[s0:s1], b.del_q[s0:s1], b.gcp[s0:s1]))
            sub = RegionBatch.from_regions([(reads, [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(nh)])])
            want = oracle.compute_batch(sub.as_dict(), n_threads=16)
            assert float(np.max(np.abs(got[r0 * nh:(r0 + 40) * nh] - want))) <= 1e-9


@pytest.mark.parametrize("name", ["config3", "config5", "ragged"])
def test_f32_first_on_the_full_sets(name):
    """The arithmetic the reference ships (pair_hmm.rs:348-366 -> gkl: f32 first, f64 where f32 cannot be trusted) on EVERY pair of
    the full-size sets, as the f64 default above: against the SIMD stand-in at the reference's own gate for that arm, 1e-5, and
    against this library's f64 results at the same gate (the f32 tests stopped at 400 / 8 regions)."""
    from lorikeet_amd import HipPairHMMEngine
    b = {"config3": synthetic.config3, "config5": synthetic.config5, "ragged": synthetic.ragged}[name]()
    e32, e64 = HipPairHMMEngine(0, f32_first=True), HipPairHMMEngine(0)
    try:
        r32 = e32.compute(b)
        assert r32.shape == (b.n_out,) and (r32 <= 0).all() and np.isfinite(r32).all()
        simd, _ = oracle.compute_batch_simd(b.as_dict(), n_threads=16, native=False)
        assert float(np.max(np.abs(r32 - simd))) <= 1e-5, "f32-first, full set vs the vector-arm stand-in"
        assert float(np.max(np.abs(r32 - e64.compute(b)))) <= 1e-5, "f32-first vs the f64 default"
    finally:
        e32.close()
        e64.close()
