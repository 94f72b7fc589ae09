"""The resident region server (lorikeet_amd/csrc/phmm_server.cpp, phmm_server_kernels.hip): phmm_region_compute /
phmm_region_submit as tasks of ONE kernel that stays on the chip -- every read of a call a wave that runs the read's whole path.
The way a private handle's region call goes once more than five handles are alive on the device (up to five keep the launched
pipeline; switch region_server = 1: every call).  Held here to
  * the launched pipeline (switch region_server = 0), field by field: everything discrete equal, likelihoods to 1e-11 (the two
    sweep a pair with different lane geometries), and to the oracle pipeline at 1e-9;
  * ITSELF, bit for bit: a region gives the same bits alone, beside other callers' regions, through the shared handle, and from
    one launch of the server to the next.
Reference: src/haplotype/haplotype_caller_engine.rs:1311-1357 (the sequence), src/assembly/assembly_region_walker.rs:210-273
(the workers that call it, one region each)."""
import threading
import time

import numpy as np
import pytest

from lorikeet_amd import region, synthetic
from lorikeet_amd.engine import HipPairHMMEngine

from project_scenarios import scenario as _scenario
from test_region_hip import _cfg, _noisy_quals, _oracle_pipeline, _priorities, _uniform_extras

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    e = HipPairHMMEngine(0)
    e.set_switch("region_server", 1)  # (every call through the server, also the lone ones a test makes)
    yield e
    e.close()


def _call(e, cfg, sc, mapq, pri, **kw):
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = sc
    return region.region_compute(e, cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, hap_priority=pri, **kw)


def _equal(a, b, exact):
    if exact:
        assert np.array_equal(a.likelihoods, b.likelihoods)
        assert np.array_equal(a.best.likelihood, b.best.likelihood) and np.array_equal(a.best.confidence, b.best.confidence, equal_nan=True)
    else:
        assert np.max(np.abs(a.likelihoods - b.likelihoods)) < 1e-11
        assert np.allclose(a.best.likelihood, b.best.likelihood, rtol=0, atol=1e-11, equal_nan=True)
    assert np.array_equal(a.keep, b.keep) and np.array_equal(a.best.allele_index, b.best.allele_index)
    assert np.array_equal(a.reads.status, b.reads.status) and np.array_equal(a.reads.new_pos, b.reads.new_pos)
    for r in range(len(a.reads.cigars)):
        assert np.array_equal(a.reads.cigars[r], b.reads.cigars[r]), r


@pytest.mark.parametrize("seed,pcr,symmetric,dynamic,low,n_regions", [(1, 3, True, False, False, 1), (2, 0, False, True, False, 3), (3, 1, True, True, True, 1),
                                                                       (4, 2, False, False, True, 5), (5, 3, True, False, False, 7)])
def test_the_server_gives_what_the_launched_pipeline_gives(eng, seed, pcr, symmetric, dynamic, low, n_regions):
    sc = _scenario(seed, n_regions=n_regions, low_complexity=low)
    b = sc[0]
    mapq = _noisy_quals(b, seed)
    cfg = _cfg(pcr=pcr, symmetric=symmetric, dynamic=dynamic)
    pri = _priorities(b, sc[1], sc[3])
    jobs = eng.stat("server_jobs")
    got = _call(eng, cfg, sc, mapq, pri)
    assert eng.stat("server_jobs") == jobs + 1, "the call did not go through the region server"
    launched = HipPairHMMEngine(0)
    try:
        launched.set_switch("region_server", 0)
        want = _call(launched, cfg, sc, mapq, pri)
        assert launched.stat("server_jobs") == jobs + 1
    finally:
        launched.close()
    _equal(got, want, exact=False)
    assert (got.reads.status == 0).sum() > b.n_reads // 3
    # ... and through the shared handle's submit / wait
    got2 = _call(eng, cfg, sc, mapq, pri, shared=True)
    assert eng.stat("server_jobs") == jobs + 2
    _equal(got2, got, exact=True)


@pytest.mark.parametrize("seed,low", [(21, False), (22, True)])
def test_the_server_gives_what_the_oracle_pipeline_gives(eng, seed, low):
    sc = _scenario(seed, n_regions=5, low_complexity=low)
    b = sc[0]
    mapq = _noisy_quals(b, seed)
    cfg = _cfg(pcr=3, dynamic=True)
    pri = _priorities(b, sc[1], sc[3])
    jobs = eng.stat("server_jobs")
    got = _call(eng, cfg, sc, mapq, pri)
    assert eng.stat("server_jobs") == jobs + 1
    out, keep, best = _oracle_pipeline(cfg, b, mapq, sc[3], pri)
    assert np.max(np.abs(got.likelihoods - out)) < 1e-9
    assert np.array_equal(got.keep, keep) and np.array_equal(got.best.allele_index, best)


def _config2_regions(n, seed):
    """n one-region calls of the bench's shape (128 reads x 8 haplotypes, 150 / 300 bases) with what the region call needs."""
    calls = []
    for i in range(n):
        b = synthetic.make_regions(1, 128, 8, 300, 150, seed=seed + i)
        extras = _uniform_extras(b, 300)
        calls.append(((b,) + tuple(extras), np.full(b.n_reads, 60, np.uint8)))
    return calls


def test_a_region_gives_the_same_bits_alone_and_beside_other_callers(eng):
    """Eight threads, an engine each, every one computing ITS region again and again while the others do the same: each call
    equals, bit for bit, what the region gave alone on an idle chip -- the forward geometry is the region's own, nothing is
    combined, nothing is planned for the load."""
    calls = _config2_regions(8, 500)
    cfg = _cfg(pcr=3)
    alone = [_call(eng, cfg, sc, mapq, None) for sc, mapq in calls]
    alone_again = [_call(eng, cfg, sc, mapq, None) for sc, mapq in calls]
    for a, b in zip(alone, alone_again):
        _equal(a, b, exact=True)
    engines = [HipPairHMMEngine(0) for _ in calls]
    for e in engines:
        e.set_switch("region_server", 1)
    errors = []

    def worker(i):
        try:
            for _ in range(12):
                _equal(_call(engines[i], cfg, calls[i][0], calls[i][1], None), alone[i], exact=True)
        except BaseException as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    jobs = eng.stat("server_jobs")
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(calls))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in engines:
        e.close()
    assert not errors, errors
    assert eng.stat("server_jobs") == jobs + 12 * len(calls)


def test_private_handles_past_five_go_through_the_server_by_default():
    """No switch set: up to five of the caller's handles on a device keep their own launched pipelines (faster there,
    profiles/r06_server_threshold.txt); with more alive -- a handle per worker thread at Lorikeet's --threads 10 -- their one-shot
    region calls go through the server, and every call gives its region's own bits whatever the others are doing."""
    calls = _config2_regions(8, 700)
    cfg = _cfg(pcr=3)
    engines = [HipPairHMMEngine(0) for _ in range(8)]
    try:
        jobs = engines[0].stat("server_jobs")
        want = [_call(engines[7], cfg, sc, mapq, None) for sc, mapq in calls]   # (eight alive: the server, one call at a time)
        assert engines[0].stat("server_jobs") == jobs + 8, "a call of one of eight private handles did not go through the server"
        launched = HipPairHMMEngine(0)
        launched.set_switch("region_server", 0)
        engines.append(launched)
        for w, (sc, mapq) in zip(want, calls):
            _equal(w, _call(launched, cfg, sc, mapq, None), exact=False)
        errors = []

        def worker(i):
            try:
                for _ in range(20):
                    _equal(_call(engines[i], cfg, calls[i][0], calls[i][1], None), want[i], exact=True)
            except BaseException as e:  # noqa: BLE001
                errors.append((i, repr(e)))

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(calls))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert engines[0].stat("server_jobs") == jobs + 8 + 20 * 8 and engines[0].stat("server_broken") == 0
    finally:
        for e in engines:
            e.close()


def test_the_server_leaves_the_chip_when_idle_and_comes_back(eng):
    sc, mapq = _config2_regions(1, 900)[0]
    cfg = _cfg(pcr=3)
    first = _call(eng, cfg, sc, mapq, None)
    launches = eng.stat("server_launches")
    time.sleep(0.05)  # (far beyond the idle time: the kernel has left)
    again = _call(eng, cfg, sc, mapq, None)
    assert eng.stat("server_launches") == launches + 1
    _equal(first, again, exact=True)
    assert eng.stat("server_broken") == 0


def test_calls_outside_the_servers_limits_take_the_launched_pipeline(eng):
    """Haplotypes beyond 512 bases (the forward instances the server carries end at 32 lanes x 16 columns): not taken, same API."""
    b = synthetic.make_regions(1, 24, 3, 600, 120, seed=77)
    extras = _uniform_extras(b, 600)
    jobs = eng.stat("server_jobs")
    got = region.region_compute(eng, _cfg(pcr=3), b, np.full(b.n_reads, 60, np.uint8), *extras)
    assert eng.stat("server_jobs") == jobs
    assert got.likelihoods.shape == (b.n_out,) and np.all(got.likelihoods <= 0)


@pytest.mark.parametrize("hap_len,n_haps", [(450, 3), (512, 5), (401, 2), (16, 4), (33, 9)])
def test_every_forward_geometry_of_the_server(eng, hap_len, n_haps):
    """16 lanes per pair up to 400 bases, 32 beyond; one to three groups of haplotypes per read: equal to the launched pipeline."""
    b = synthetic.make_regions(2, 20, n_haps, hap_len, min(120, hap_len), seed=hap_len)
    extras = _uniform_extras(b, hap_len)
    mapq = np.full(b.n_reads, 60, np.uint8)
    jobs = eng.stat("server_jobs")
    got = region.region_compute(eng, _cfg(pcr=3), b, mapq, *extras)
    assert eng.stat("server_jobs") == jobs + 1
    launched = HipPairHMMEngine(0)
    try:
        launched.set_switch("region_server", 0)
        want = region.region_compute(launched, _cfg(pcr=3), b, mapq, *extras)
    finally:
        launched.close()
    _equal(got, want, exact=False)


def test_two_tickets_per_thread_on_the_shared_handle(eng):
    """phmm_region_submit twice, then phmm_wait twice (tools/threads_bench TB_DEPTH=2): both through the server, both right."""
    import ctypes as C
    from lorikeet_amd import _lib
    calls = _config2_regions(2, 1200)
    cfg = _cfg(pcr=3)
    want = [_call(eng, cfg, sc, mapq, None) for sc, mapq in calls]
    jobs = eng.stat("server_jobs")
    results, errors = [None, None], []

    def worker(i):
        try:
            results[i] = _call(eng, cfg, calls[i][0], calls[i][1], None, shared=True)
        except BaseException as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert eng.stat("server_jobs") == jobs + 2
    for got, w in zip(results, want):
        _equal(got, w, exact=True)
    assert C.sizeof(_lib.EngineConfig) > 0


_LOST_ANSWER = r"""
import numpy as np
from lorikeet_amd import region
from lorikeet_amd.engine import HipPairHMMEngine
from test_server_hip import _call, _config2_regions, _equal
from test_region_hip import _cfg
(sc, mapq), (sc2, mapq2) = _config2_regions(2, 4100)
cfg = _cfg(pcr=3)
launched = HipPairHMMEngine(0)
launched.set_switch("region_server", 0)
want, want2 = _call(launched, cfg, sc, mapq, None), _call(launched, cfg, sc2, mapq2, None)
e = HipPairHMMEngine(0)
e.set_switch("region_server", 1)
ok = _call(e, cfg, sc, mapq, None)
assert e.stat("server_jobs") == 1 and e.stat("server_broken") == 0
_equal(ok, want, exact=False)
e.set_switch("region_debug_pick", 4)          # the next answer counts as lost: the call is run again by the launched pipeline
got = _call(e, cfg, sc2, mapq2, None)
assert e.stat("server_jobs") == 2 and e.stat("server_broken") == 1
_equal(got, want2, exact=True)                # (the launched pipeline's own bits)
e.set_switch("region_debug_pick", 0)
again = _call(e, cfg, sc, mapq, None)         # ... and so is every later call of the process: the server is not used again
assert e.stat("server_jobs") == 2
_equal(again, want, exact=True)
others = [HipPairHMMEngine(0) for _ in range(8)]
_equal(_call(others[7], cfg, sc2, mapq2, None), want2, exact=True)
assert e.stat("server_jobs") == 2
print("fallback ok")
"""


def test_a_lost_answer_sends_the_call_and_all_later_ones_to_the_launched_pipeline():
    """The server's own failure never fails a call: a stall (nothing finishing for 5 s) or a call that does not come back marks the
    device's server broken -- phmm_stat "server_broken" --, the call is run again by the launched pipeline and so is every later
    one.  In a process of its own (the mark stays for the process' life); the lost answer is the test hook region_debug_pick & 4."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]), TMPDIR="/tmp")
    r = subprocess.run([sys.executable, "-c", _LOST_ANSWER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_output_slots_too_small_are_reported_by_the_server_like_by_the_launched_pipeline(eng):
    """capacity = 1: the projected CIGARs do not fit the caller's slots -- PHMM_ERR_CIGAR_CAPACITY with the sizes in n_out_cigar,
    the mirror retries with those; both attempts go through the server and the result is the launched pipeline's."""
    sc = _scenario(31, n_regions=3, low_complexity=False)
    b = sc[0]
    mapq = _noisy_quals(b, 31)
    cfg = _cfg(pcr=3)
    pri = _priorities(b, sc[1], sc[3])
    jobs = eng.stat("server_jobs")
    got = _call(eng, cfg, sc, mapq, pri, capacity=1)
    assert eng.stat("server_jobs") == jobs + 2, "capacity 1 must fail once (sizes reported) and pass on the retry, both through the server"
    launched = HipPairHMMEngine(0)
    try:
        launched.set_switch("region_server", 0)
        _equal(got, _call(launched, cfg, sc, mapq, pri), exact=False)
    finally:
        launched.close()
    assert max(len(c) for c in got.reads.cigars) > 1


def test_a_handles_flags_hold_inside_the_server():
    """PHMM_FLAG_NO_TRISTATE: the server sweeps with the HANDLE'S tables (mismatch prior eps instead of eps / 3), so its results are
    that handle's launched results -- and differ from a default handle's.  PHMM_FLAG_F32_FIRST: the server has one arithmetic, f64
    (what the f32-first mode falls back to): such a handle's calls through the server give the f64 results."""
    sc = _scenario(41, n_regions=2, low_complexity=False)
    b = sc[0]
    mapq = _noisy_quals(b, 41)
    cfg = _cfg(pcr=3)
    pri = _priorities(b, sc[1], sc[3])
    plain = HipPairHMMEngine(0)
    plain.set_switch("region_server", 0)
    want_f64 = _call(plain, cfg, sc, mapq, pri)
    for kw, same_as_plain in (({"do_not_use_tristate_correction": True}, False), ({"f32_first": True}, True)):
        served, launched = HipPairHMMEngine(0, **kw), HipPairHMMEngine(0, **kw)
        try:
            served.set_switch("region_server", 1)
            launched.set_switch("region_server", 0)
            jobs = served.stat("server_jobs")
            got = _call(served, cfg, sc, mapq, pri)
            assert served.stat("server_jobs") == jobs + 1
            if same_as_plain:
                _equal(got, want_f64, exact=False)            # f64, whatever the launched f32-first sweep rounds to
                assert np.max(np.abs(_call(launched, cfg, sc, mapq, pri).likelihoods - got.likelihoods)) < 1e-5
            else:
                _equal(got, _call(launched, cfg, sc, mapq, pri), exact=False)
                assert np.max(np.abs(got.likelihoods - want_f64.likelihoods)) > 1e-3
        finally:
            served.close()
            launched.close()
    plain.close()
