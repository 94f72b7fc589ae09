"""phmm_compute_multi: one process, several engines, whole regions sharded by cells (contiguous ranges, greedy LPT for
heavy-tailed sets) and nothing exchanged between them.  On a node with several GPUs the engines sit on
DISTINCT devices (`_engines`); on a one-GPU box the same tests exercise the sharding and staging logic with several
engines on that device -- they say nothing about several devices, and `test_one_engine_per_device` is skipped there
rather than aliased."""
import numpy as np
import pytest

from lorikeet_amd import HipPairHMMEngine, PhmmError, synthetic
from lorikeet_amd import _lib
from lorikeet_amd.batch import Read, RegionBatch
from lorikeet_amd.engine import assign_regions, compute_multi, split_regions
from oracle import oracle
from test_sharding import _ragged_batch

pytestmark = pytest.mark.gpu


def _device_count():
    return int(_lib.load().phmm_device_count())


def _engines(n):
    """n engines: on distinct devices where the node has that many, else as many devices as there are, round robin."""
    nd = max(_device_count(), 1)
    return [HipPairHMMEngine(i % nd) for i in range(n)]


def test_one_engine_per_device():
    """BASELINE.json configs[3] inside ONE process: the engines of phmm_compute_multi on distinct devices (each with its own
    host thread pinned to the CPUs local to its GPU).  Skipped -- not aliased onto one device -- on a one-GPU box."""
    nd = _device_count()
    if nd < 2:
        pytest.skip("one HIP device visible: nothing to shard across")
    engines = [HipPairHMMEngine(d) for d in range(min(nd, 8))]
    for b in (synthetic.config3(64 * len(engines), seed=21),
              RegionBatch.concat([synthetic.make_regions(1, 1200, 8, 300, 150, seed=5), synthetic.config2(3 * len(engines), seed=6)])):
        before = [e.stat("staged_bytes") for e in engines]
        got = compute_multi(engines, b)
        staged = [e.stat("staged_bytes") - x for e, x in zip(engines, before)]
        assert all(x > 0 for x in staged) and sum(staged) == 5 * int(b.read_off[-1]) + int(b.hap_off[-1]), staged
        assert np.max(np.abs(got - engines[0].compute(b))) <= 1e-12      # every device computes what device 0 computes
        assert np.max(np.abs(got - engines[-1].compute(b))) <= 1e-12
    # ... and the whole per-region path (phmm_region_compute_multi): every device's share equal to what device 0 makes of it
    # (phmm_engine_compute_multi runs the same ranges through the engine-level call: test_engine_call_over_several_engines)
    _region_call_over(engines, n_regions=4 * len(engines))
    for e in engines:
        e.close()


def test_first_call_on_a_handle_of_another_device_than_the_threads_current_one():
    """the entry points take the handle's own hardware queue (latch_slot0 -> queues_acquire) before their
    device guard, and hipExtStreamCreateWithCUMask creates on the calling thread's CURRENT device.  A thread that sits on device 0
    -- every fresh worker thread does -- making the FIRST call on a handle of device 1 must get queues of device 1: its
    one-region calls (phmm_compute, phmm_region_compute, all-pairs and chain) equal device 0's."""
    nd = _device_count()
    if nd < 2:
        pytest.skip("one HIP device visible")
    import ctypes as C
    import threading
    from lorikeet_amd import region
    from project_scenarios import scenario
    from test_region_hip import _cfg, _equal_calls, _noisy_quals
    hip = C.CDLL("libamdhip64.so")
    b = synthetic.make_regions(1, 24, 4, 200, [60, 100], seed=77)
    sc = scenario(78, n_regions=1)
    mapq = _noisy_quals(sc[0], 78)
    e0 = HipPairHMMEngine(0)
    want_lk, want_region = e0.compute(b), region.region_compute(e0, _cfg(), sc[0], mapq, *sc[1:])
    result = {}

    def worker():  # a fresh thread: current device 0
        assert hip.hipSetDevice(0) == 0
        e1 = HipPairHMMEngine(1)
        try:
            result["lk"] = e1.compute(b)                    # the FIRST one-enqueue call latches the queue pair
            for sw_all in (1 << 20, 0):
                e1.set_switch("region_sw_all", sw_all)
                result[sw_all] = region.region_compute(e1, _cfg(), sc[0], mapq, *sc[1:])
            dev = C.c_int(-1)
            hip.hipGetDevice(C.byref(dev))
            result["device_after"] = dev.value
        finally:
            e1.close()

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert np.array_equal(result["lk"], want_lk)
    _equal_calls(result[1 << 20], want_region)
    _equal_calls(result[0], want_region)
    assert result["device_after"] == 0                      # the guards put the thread's device back
    e0.close()


def _region_call_over(engines, n_regions, seed=5):
    """phmm_region_compute_multi over `engines` == phmm_region_compute on the first of them, field by field."""
    from lorikeet_amd import region
    from project_scenarios import scenario
    from test_region_hip import _cfg, _equal_calls, _noisy_quals, _priorities
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = scenario(seed, n_regions=n_regions)
    mapq = _noisy_quals(b, seed)
    pri = _priorities(b, hap_cigars, ref_hap)
    cfg = _cfg()
    call = lambda **kw: region.region_compute(engines[0], cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars,  # noqa: E731
                                              hap_priority=pri, **kw)
    want = call()
    got = call(engines=engines)
    _equal_calls(got, want)
    assert (want.reads.status == 0).sum() >= 1
    return b


def test_region_call_over_several_engines():
    """the per-region call has its multi-engine entry point too.  (Engines on distinct devices where the
    node has them; on a one-GPU box this exercises the splitting, the rebased ranges and the threads.)"""
    engines = _engines(3)
    b = _region_call_over(engines, n_regions=11)
    _region_call_over(engines[:1], n_regions=3, seed=6)       # one engine: plain phmm_region_compute
    _region_call_over(engines, n_regions=2, seed=7)           # fewer regions than engines
    # argument errors are caught once, for the whole call
    from lorikeet_amd import region
    from project_scenarios import scenario
    from test_region_hip import _cfg
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = scenario(8, n_regions=4)
    with pytest.raises(PhmmError) as e:
        region.region_compute(engines[0], _cfg(pcr=9), b, np.full(b.n_reads, 60, np.uint8), hap_cigars, hap_starts, ref_hap, ref_start,
                              orig_cigars, engines=engines)
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG and "PCR" in str(e.value)
    for x in engines:
        x.close()


def test_engine_call_over_several_engines():
    """phmm_engine_compute_multi == phmm_engine_compute on one engine: normalised likelihoods and keep flags, with and without
    the indel quality tracks, fewer regions than engines, errors once for the whole call."""
    import ctypes as C
    from test_region_hip import _cfg, _engine_compute, _noisy_quals
    from project_scenarios import scenario
    engines = _engines(3)
    hs = (C.c_void_p * 3)(*[e._h for e in engines])
    pp = lambda a, t: None if a is None else a.ctypes.data_as(t)  # noqa: E731
    for n_regions, tags in ((13, True), (2, False)):
        b, _, _, ref_hap, _, _ = scenario(40 + n_regions, n_regions=n_regions)
        mapq = _noisy_quals(b, n_regions)
        cfg = _cfg(pcr=3, dynamic=True)
        if not tags:
            b.ins_q = b.del_q = None
        out, keep = np.full(b.n_out, np.nan), np.zeros(b.n_reads, np.uint8)
        rr = np.ascontiguousarray(ref_hap, np.int32)
        args = (C.byref(cfg), b.n_regions, pp(b.region_read_off, _lib.u32p), pp(b.region_hap_off, _lib.u32p), pp(b.read_off, _lib.u32p),
                pp(b.read_bases, _lib.u8p), pp(b.base_q, _lib.u8p), pp(b.ins_q, _lib.u8p), pp(b.del_q, _lib.u8p), pp(mapq, _lib.u8p),
                pp(b.hap_off, _lib.u32p), pp(b.hap_bases, _lib.u8p), pp(rr, C.POINTER(C.c_int32)), pp(b.out_off, _lib.u64p))
        st = engines[0].lib.phmm_engine_compute_multi(hs, 3, *args, pp(out, _lib.f64p), pp(keep, _lib.u8p))
        assert st == 0, engines[0].last_error()
        want, wkeep = np.full(b.n_out, np.nan), np.zeros(b.n_reads, np.uint8)
        st = engines[0].lib.phmm_engine_compute(engines[0]._h, *args, pp(want, _lib.f64p), pp(wkeep, _lib.u8p))
        assert st == 0, engines[0].last_error()
        assert np.max(np.abs(out - want)) <= 1e-12 and np.array_equal(keep, wkeep)
    bad = _cfg(pcr=9)
    st = engines[0].lib.phmm_engine_compute_multi(hs, 3, C.byref(bad), *args[1:], pp(out, _lib.f64p), pp(keep, _lib.u8p))
    assert st == _lib.PHMM_ERR_INVALID_ARG and "PCR" in engines[0].last_error()
    for e in engines:
        e.close()


def test_several_engines_one_call():
    engines = _engines(3)
    ragged = _ragged_batch()            # regions without reads, 1-4 haplotypes, very different cell counts
    want = oracle.compute_batch(ragged.as_dict(), n_threads=4)
    got = compute_multi(engines, ragged)
    assert np.max(np.abs(got - want)) <= 1e-9
    for b in (synthetic.config3(40, seed=8), synthetic.config2(2, seed=9)):   # the second: fewer regions than engines
        single = engines[0].compute(b)
        assert np.max(np.abs(compute_multi(engines, b) - single)) <= 1e-12
        assert np.max(np.abs(compute_multi(engines[:1], b) - single)) == 0.0   # one engine: plain phmm_compute
    part = assign_regions(ragged, 3)
    assert set(part.tolist()) <= {0, 1, 2} and len(part) == ragged.n_regions
    for e in engines:
        e.close()


def test_errors_of_a_share_reach_the_caller():
    engines = _engines(2)
    hap = np.full(40, ord("A"), np.uint8)
    n = 8
    weird = ([Read(hap[:n].copy(), np.full(n, 93), np.zeros(n, int), np.zeros(n, int), np.full(n, 10))], [hap])
    plain = ([Read(hap[:20].copy(), np.full(20, 30), np.full(20, 40), np.full(20, 40), np.full(20, 10))], [hap])
    b = RegionBatch.from_regions([weird, plain, plain, plain])
    with pytest.raises(PhmmError) as e:
        compute_multi(engines, b)
    assert e.value.code == _lib.PHMM_ERR_POSITIVE_RESULT and "greater than 0.0" in str(e.value)
    good = synthetic.config2(3, seed=4)
    assert np.max(np.abs(compute_multi(engines, good) - engines[0].compute(good))) <= 1e-12   # the handles stay usable
    bad = synthetic.make_regions(2, 4, 2, 100, 50, seed=1)
    bad.read_off = bad.read_off.copy()
    bad.read_off[2] = bad.read_off[1] - 1
    with pytest.raises(PhmmError) as e:
        compute_multi(engines, bad)
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG and "monotonic" in str(e.value)
    for x in engines:
        x.close()


def test_no_payload_byte_is_copied_more_than_once():
    """phmm_compute_multi used to gather every engine's share on the calling thread and then copy it again
    into the pinned mirror.  Now every engine stages straight from the caller's arrays: the engines' staging counters add
    up to exactly the payload (5 bytes per read base + 1 per haplotype base), in both assignment modes."""
    engines = _engines(3)
    uniform = synthetic.config3(90, seed=12)                    # contiguous cell-balanced ranges, chunked per engine
    heavy = RegionBatch.concat([synthetic.make_regions(1, 1200, 8, 300, 150, seed=5), synthetic.config2(6, seed=6),
                                synthetic.make_regions(12, 10, 2, 80, 40, seed=7)])   # one region dominates: LPT lists
    cells = lambda b: [int(c) for c in __import__("lorikeet_amd").sharding.region_cells(b)]  # noqa: E731
    from lorikeet_amd import sharding
    assert sharding.imbalance(cells(uniform), bounds=split_regions(uniform, 3).tolist()) <= 1.05
    assert sharding.imbalance(cells(heavy), bounds=split_regions(heavy, 3).tolist()) > 1.05
    for b in (uniform, heavy):
        before = sum(e.stat("staged_bytes") for e in engines)
        got = compute_multi(engines, b)
        staged = sum(e.stat("staged_bytes") for e in engines) - before
        assert staged == 5 * int(b.read_off[-1]) + int(b.hap_off[-1]), (staged, b.algorithmic_bytes())
        assert np.max(np.abs(got - engines[0].compute(b))) <= 1e-12
        sub = b.region_slice(0, 2)
        assert np.max(np.abs(got[:sub.n_out] - oracle.compute_batch(sub.as_dict(), n_threads=8))) <= 1e-9
    # gaps in out_off survive both modes
    need = np.diff(heavy.out_off.astype(np.int64))
    off = np.concatenate([[0], np.cumsum(need + 2)]).astype(np.uint64)
    d = heavy.as_dict()
    d["out_off"] = off
    gapped = RegionBatch(**d)
    import ctypes as C
    out = np.full(int(off[-1]), -7.5)
    hs = (C.c_void_p * 3)(*[e._h for e in engines])
    st = engines[0].lib.phmm_compute_multi(hs, 3, *HipPairHMMEngine._abi_args(gapped), out.ctypes.data_as(_lib.f64p))
    assert st == 0, engines[0].last_error()
    want = engines[0].compute(heavy)
    for g in range(heavy.n_regions):
        o, n = int(off[g]), int(need[g])
        assert np.max(np.abs(out[o:o + n] - want[int(heavy.out_off[g]):int(heavy.out_off[g + 1])])) <= 1e-12
        assert np.all(out[o + n:int(off[g + 1])] == -7.5)
    for e in engines:
        e.close()
