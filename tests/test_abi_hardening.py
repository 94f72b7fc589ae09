"""Boundary behaviour the reviews of round 1 asked to pin down (VERDICT 'weak', ADVICE): an empty haplotype is an
explicit error at every entry point, whole-batch validation happens before the chunked path indexes anything, gaps in
`out` stay untouched, developer switches are per handle."""
import ctypes as C

import numpy as np
import pytest

from lorikeet_amd import HipPairHMMEngine, PhmmError, _lib, synthetic
from lorikeet_amd.batch import Read, RegionBatch
from lorikeet_amd.engine import compute_multi
from oracle import oracle

pytestmark = pytest.mark.gpu


def _with_empty_haplotype(n_regions):
    b = synthetic.make_regions(n_regions, 6, 3, 50, 30, seed=4)
    d = b.as_dict()
    # haplotype 1 of the last region loses its bases: hap_off[k+1] == hap_off[k]
    k = int(b.region_hap_off[-2]) + 1
    ho = b.hap_off.astype(np.int64).copy()
    ln = ho[k + 1] - ho[k]
    d["hap_bases"] = np.concatenate([b.hap_bases[:ho[k]], b.hap_bases[ho[k + 1]:]])
    ho[k + 1:] -= ln
    d["hap_off"] = ho.astype(np.uint32)
    return RegionBatch(**d)


@pytest.mark.parametrize("n_regions", [1, 40, 900])  # one shot, shared launch, chunked pipeline (> 0.5 MB per array)
def test_empty_haplotype_is_rejected_by_every_entry_point(hip_engine, n_regions):
    """The reference would compute 2^1020 / 0 and return -inf for the whole region (pair_hmm.rs:515-517); SURVEY 8b asks
    for an explicit error instead.  PHMM_ERR_INVALID_ARG, message names the cause, the engine stays usable."""
    bad = _with_empty_haplotype(n_regions)
    with pytest.raises(PhmmError) as ei:
        hip_engine.compute(bad)
    assert ei.value.code == _lib.PHMM_ERR_INVALID_ARG and "empty haplotype" in str(ei.value)
    with pytest.raises(PhmmError, match="empty haplotype"):
        hip_engine.plan(bad)
    with pytest.raises(PhmmError, match="empty haplotype"):
        hip_engine.submit(bad)
    e2 = HipPairHMMEngine(0)
    with pytest.raises(PhmmError, match="empty haplotype"):
        compute_multi([hip_engine, e2], bad)
    e2.close()
    # engine-level call
    cfg = _lib.EngineConfig()
    cfg.constant_gcp, cfg.base_quality_score_threshold = 10, 18
    cfg.log10_global_read_mismapping_rate = -4.5
    mapq = np.full(bad.n_reads, 60, np.uint8)
    out = np.empty(bad.n_out, np.float64)
    keep = np.zeros(bad.n_reads, np.uint8)
    pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
    st = hip_engine.lib.phmm_engine_compute(
        hip_engine._h, C.byref(cfg), bad.n_regions, pp(bad.region_read_off, _lib.u32p), pp(bad.region_hap_off, _lib.u32p),
        pp(bad.read_off, _lib.u32p), pp(bad.read_bases, _lib.u8p), pp(bad.base_q, _lib.u8p), None, None, pp(mapq, _lib.u8p),
        pp(bad.hap_off, _lib.u32p), pp(bad.hap_bases, _lib.u8p), None, pp(bad.out_off, _lib.u64p), pp(out, _lib.f64p),
        pp(keep, _lib.u8p))
    assert st == _lib.PHMM_ERR_INVALID_ARG and "empty haplotype" in hip_engine.last_error()
    ok = synthetic.make_regions(2, 4, 2, 40, 20, seed=5)
    assert np.max(np.abs(hip_engine.compute(ok) - oracle.compute_batch(ok.as_dict()))) < 1e-9


def test_a_region_without_haplotypes_or_reads_is_still_fine(hip_engine):
    rd = Read(b"ACGTAC", [30] * 6, [40] * 6, [40] * 6, [10] * 6)
    b = RegionBatch.from_regions([([rd], []), ([], [b"ACGTACGT"]), ([rd], [b"ACGTACGT"])])
    got = hip_engine.compute(b)
    assert got.shape == (1,) and abs(got[0] - oracle.compute_batch(b.as_dict())[0]) < 1e-9


def test_large_batches_are_validated_before_anything_is_indexed(hip_engine):
    """the chunked path (>= 8 regions, > 0.5 MB per array) used to walk the caller's offset arrays
    before any check; garbage then meant out-of-bounds reads or a multi-GB resize whose bad_alloc crossed the C ABI."""
    b = synthetic.config2(64, seed=8)
    assert b.read_off[-1] > (512 << 10)
    for key, mutate in (("region_read_off", lambda a: a[::-1].copy()),
                        ("region_hap_off", lambda a: np.where(np.arange(len(a)) == 5, 0xfffffff0, a).astype(np.uint32)),
                        ("read_off", lambda a: np.where(np.arange(len(a)) == 4000, 7, a).astype(np.uint32)),
                        ("hap_off", lambda a: np.where(np.arange(len(a)) == 17, 3, a).astype(np.uint32)),
                        ("out_off", lambda a: (a // 2).astype(np.uint64))):
        d = b.as_dict()
        d[key] = mutate(getattr(b, key))
        d[key][0] = 0
        bad = RegionBatch(**d)
        with pytest.raises(PhmmError) as ei:
            hip_engine.compute(bad)
        assert ei.value.code == _lib.PHMM_ERR_INVALID_ARG, key
    # null payload pointer with a non-empty batch
    args = list(HipPairHMMEngine._abi_args(b))
    out = np.empty(b.n_out, np.float64)
    args[6] = None  # ins_q
    st = hip_engine.lib.phmm_compute(hip_engine._h, *args, out.ctypes.data_as(_lib.f64p))
    assert st == _lib.PHMM_ERR_INVALID_ARG and "null pointer" in hip_engine.last_error()
    assert np.max(np.abs(hip_engine.compute(b)[:1024] - oracle.compute_batch(b.region_slice(0, 1).as_dict()))) < 1e-9


@pytest.mark.parametrize("n_regions", [3, 60, 700])
def test_gaps_in_out_stay_untouched(hip_engine, n_regions):
    """out_off may leave room between regions; those slots belong to the caller (they used to come back NaN)."""
    b = synthetic.make_regions(n_regions, 20, 3, 60, [25, 40], seed=6)
    want = hip_engine.compute(b)
    need = np.diff(b.out_off.astype(np.int64))
    pad = np.arange(n_regions) % 3 + 1                       # 1..3 spare slots behind every region
    off = np.concatenate([[0], np.cumsum(need + pad)]).astype(np.uint64)
    d = b.as_dict()
    d["out_off"] = off
    gb = RegionBatch(**d)
    sentinel = -12345.678
    for route in ("compute", "submit"):
        out = np.full(int(off[-1]), sentinel)
        args = HipPairHMMEngine._abi_args(gb)
        if route == "compute":
            assert hip_engine.lib.phmm_compute(hip_engine._h, *args, out.ctypes.data_as(_lib.f64p)) == 0, hip_engine.last_error()
        else:
            t = C.c_uint64(0)
            assert hip_engine.lib.phmm_submit(hip_engine._h, *args, out.ctypes.data_as(_lib.f64p), C.byref(t)) == 0
            assert hip_engine.lib.phmm_wait(hip_engine._h, t.value) == 0
        for g in range(n_regions):
            o, n = int(off[g]), int(need[g])
            assert np.array_equal(out[o:o + n], want[int(b.out_off[g]):int(b.out_off[g + 1])]), (route, g)
            assert np.all(out[o + n:int(off[g + 1])] == sentinel), (route, g)


def test_switches_are_per_handle_and_unknown_names_are_refused(hip_engine):
    b = synthetic.config2(24, seed=9)
    other = HipPairHMMEngine(0)
    other.set_switch("force_chain", 16)
    other.set_switch("force_L", 16)
    p0, p1 = hip_engine.plan(b), other.plan(b)
    assert p1.dominant_kernel.replace("chain_k<", "chain<").startswith("phmm_forward_chain<16,") and not p0.dominant_kernel.startswith("phmm_forward_chain")
    p0.close()
    p1.close()
    with pytest.raises(PhmmError):
        other.set_switch("no_such_switch", 1)
    other.close()
