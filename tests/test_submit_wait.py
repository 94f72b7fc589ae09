"""phmm_submit / phmm_wait: the reference's call pattern -- every rayon worker calls PairHMM::compute_likelihoods with
one region at a time (pair_hmm.rs:345-375 from assembly_region_walker.rs:210-273) -- served by ONE shared handle that
computes whatever regions are waiting as one batch.  Every submission must get exactly its own results and its own
status, whichever thread led the flush it travelled in."""
import threading

import numpy as np
import pytest

from lorikeet_amd import HipPairHMMEngine, PhmmError, synthetic
from lorikeet_amd import _lib
from lorikeet_amd.batch import Read, RegionBatch
from oracle import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _regions(n, seed):
    """n one-region batches of different shapes (reads x haplotypes, lengths), with the oracle's answer for each."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        b = synthetic.make_regions(1, int(rng.integers(1, 40)), int(rng.integers(1, 9)), int(rng.integers(120, 420)),
                                   int(rng.integers(30, 120)), seed=seed * 1000 + i)
        out.append((b, oracle.compute_batch(b.as_dict(), n_threads=4)))
    return out


def test_many_threads_share_one_handle():
    eng = HipPairHMMEngine(0)
    T, per_thread = 8, 12
    work = [_regions(per_thread, 50 + t) for t in range(T)]
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for b, want in work[t]:
                    ticket, out = eng.submit(b)
                    eng.wait(ticket)
                    d = float(np.max(np.abs(out - want)))
                    assert d <= TOL, (t, d)
        except Exception as e:  # surfaced in the main thread
            errors.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[0]
    flushes, subs = eng.submit_stats()
    assert subs == T * per_thread * 3
    assert 1 <= flushes <= subs
    eng.close()


def test_one_thread_many_tickets_any_wait_order():
    eng = HipPairHMMEngine(0)
    items = _regions(9, 77)
    tickets = [eng.submit(b) for b, _ in items]
    # the first wait flushes everything queued; the others only collect
    for i in (4, 0, 8, 1, 2, 3, 7, 6, 5):
        eng.wait(tickets[i][0])
        assert np.max(np.abs(tickets[i][1] - items[i][1])) <= TOL
    flushes, subs = eng.submit_stats()
    assert (flushes, subs) == (1, 9)
    # the same regions through phmm_compute give the same numbers (which regions share a launch is invisible)
    own = HipPairHMMEngine(0)
    for (b, _), (_, out) in zip(items, tickets):
        assert np.max(np.abs(own.compute(b) - out)) <= 1e-12
    own.close()
    with pytest.raises(PhmmError) as e:  # a ticket is good for one wait
        eng.wait(tickets[0][0])
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG and "ticket" in str(e.value)
    eng.close()


def test_multi_region_and_large_submissions_mix():
    """Submissions may hold many regions; one that is too large to share a staging pass is computed on its own."""
    eng = HipPairHMMEngine(0)
    small = [synthetic.config2(3, seed=5), synthetic.config3(2, seed=6)]
    big = synthetic.config2(300, seed=7)  # 5.8 MB per per-base array: more than one combined staging pass takes
    subs = [eng.submit(b) for b in (small[0], big, small[1])]
    for t, _ in subs:
        eng.wait(t)
    own = HipPairHMMEngine(0)
    for b, (_, out) in zip((small[0], big, small[1]), subs):
        assert np.all(out <= 0.0)
        assert np.max(np.abs(own.compute(b) - out)) <= 1e-12
    want = oracle.compute_batch(small[1].as_dict(), n_threads=8)
    assert np.max(np.abs(subs[2][1] - want)) <= TOL
    own.close()
    eng.close()


def test_argument_errors_stay_with_their_submitter():
    eng = HipPairHMMEngine(0)
    good = _regions(3, 91)
    t0, out0 = eng.submit(good[0][0])
    bad = synthetic.make_regions(1, 4, 2, 100, 50, seed=1)
    bad.read_off = bad.read_off.copy()
    bad.read_off[2] = bad.read_off[1] - 1  # not monotonic
    bad.__dict__.pop("_abi_args", None)
    with pytest.raises(PhmmError) as e:
        eng.submit(bad)
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG and "monotonic" in str(e.value)
    t1, out1 = eng.submit(good[1][0])
    eng.wait(t1)
    eng.wait(t0)
    assert np.max(np.abs(out0 - good[0][1])) <= TOL and np.max(np.abs(out1 - good[1][1])) <= TOL
    # empty submissions (no regions / a region without reads) are legal, as in phmm_compute
    empty = RegionBatch.from_regions([([], [np.frombuffer(b"ACGT", np.uint8)])])
    t, out = eng.submit(empty)
    eng.wait(t)
    assert out.size == 0
    eng.close()


def test_positive_result_is_reported_to_its_owner_only():
    """A region that trips the reference's `<= 0` assert (pair_hmm.rs:478-481; reachable with gap-open qualities below 3,
    where the transitions out of the match state sum to more than one) fails for its submitter -- for the oracle as for
    phmm_compute -- and the regions that shared its flush are served."""
    hap = np.full(40, ord("A"), np.uint8)
    n = 8
    weird = RegionBatch.from_regions([([Read(hap[:n].copy(), np.full(n, 93), np.zeros(n, int), np.zeros(n, int), np.full(n, 10))],
                                       [hap])])
    with pytest.raises(AssertionError):
        oracle.compute_batch(weird.as_dict(), n_threads=1)
    own = HipPairHMMEngine(0)
    with pytest.raises(PhmmError) as e:
        own.compute(weird)
    assert e.value.code == _lib.PHMM_ERR_POSITIVE_RESULT
    own.close()
    eng = HipPairHMMEngine(0)
    good = _regions(4, 123)
    tickets = [eng.submit(good[0][0]), eng.submit(weird), eng.submit(good[1][0]), eng.submit(good[2][0])]
    with pytest.raises(PhmmError) as e:
        eng.wait(tickets[1][0])
    assert e.value.code == _lib.PHMM_ERR_POSITIVE_RESULT and "greater than 0.0" in str(e.value)
    for i, g in ((0, 0), (2, 1), (3, 2)):
        eng.wait(tickets[i][0])
        assert np.max(np.abs(tickets[i][1] - good[g][1])) <= TOL
    assert eng.submit_stats() == (1, 4)
    eng.close()


def test_engine_level_submissions_share_flushes():
    """phmm_engine_submit: worker threads with an engine object each (as the reference clones its engine per task) on
    one shared handle.  Regions of equal configuration are computed together; two configurations and plain phmm_submit
    traffic on the same handle are kept apart and every caller gets the results of a private engine."""
    import math
    from lorikeet_amd.likelihood_engine import PairHMMLikelihoodCalculationEngine, PCRErrorModel
    from test_engine_hip import _random_regions

    shared = HipPairHMMEngine(0)
    cap = -4.5 * math.log10(math.e)
    confs = [(10, cap, PCRErrorModel.CONSERVATIVE, 18, True, 1.0, 0.02, True, False),
             (10, cap, PCRErrorModel.NONE, 18, False, 1.0, 0.02, False, False)]
    T = 6
    work = [_random_regions(np.random.default_rng(900 + t), 10, with_tags=bool(t % 2)) for t in range(T)]
    want = []
    for t in range(T):
        private = PairHMMLikelihoodCalculationEngine(*confs[t % 2])
        want.append([private.compute_regions([reg])[0] for reg in work[t]])
    plain = _regions(6, 321)
    errors = []

    def worker(t):
        try:
            eng = PairHMMLikelihoodCalculationEngine(*confs[t % 2], shared_engine=shared)
            for rep in range(2):
                for reg, (wm, wk) in zip(work[t], want[t]):
                    (gm, gk), = eng.compute_regions([reg])
                    assert gm.shape == wm.shape and np.array_equal(gk, wk)
                    if gm.size:
                        assert np.max(np.abs(gm - wm)) <= 1e-12
        except Exception as e:
            errors.append(e)

    def plain_worker():
        try:
            for rep in range(4):
                for b, w in plain:
                    ticket, out = shared.submit(b)
                    shared.wait(ticket)
                    assert np.max(np.abs(out - w)) <= TOL
        except Exception as e:
            errors.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)] + [threading.Thread(target=plain_worker)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[0]
    flushes, subs = shared.submit_stats()
    assert subs == T * 10 * 2 + 4 * 6 and flushes <= subs
    shared.close()


def test_shared_handle_in_f32_first_mode():
    """The lanes of a shared handle inherit its flags: submissions that together are large enough are swept in f32 first
    (gate 1e-5, north_star), small ones are f64 as on a private f32-first handle."""
    shared = HipPairHMMEngine(0, f32_first=True)
    e64 = HipPairHMMEngine(0)
    # 3 x 60 regions: none is large enough for the chained (f32) kernel on its own, the 182 regions of the flush are
    big = [synthetic.config2(60, seed=70 + i) for i in range(3)]
    small = synthetic.config2(2, seed=75)
    tickets = [shared.submit(b) for b in big] + [shared.submit(small)]
    for t, _ in tickets:
        shared.wait(t)
    assert shared.submit_stats() == (1, 4)
    worst = 0.0
    for b, (_, out) in zip(big + [small], tickets):
        ref = e64.compute(b)
        assert np.all(out <= 0.0) and not np.isnan(out).any()
        worst = max(worst, float(np.max(np.abs(out - ref))))
    assert 0.0 < worst <= 1e-5, worst   # f32 results: not the f64 numbers, inside the gate
    want = oracle.compute_batch(small.as_dict(), n_threads=4)
    assert np.max(np.abs(tickets[3][1] - want)) <= 1e-5
    e64.close()
    shared.close()


def test_many_private_handles_are_served_by_the_devices_shared_lanes():
    """INTEGRATION.md section 4 gives every rayon worker a handle of its own (thread_local!).  Past four of the caller's handles
    alive on a device their one-shot calls go through ONE shared handle inside the library (phmm_host::route_shared: phmm_submit /
    phmm_wait, the callers that are waiting anyway merge into one flush) when PHMM_ROUTE_SHARED asks for it -- 32 private handles
    used to run at HALF the rate of 16.  Same results, per handle, as a lone handle's (to 1e-11: a combined flush may sweep a
    region with another lane geometry, which is why the routing is opt-in); a handle whose switches were touched stays on its own
    streams."""
    lone = HipPairHMMEngine(0)
    work = _regions(10, 91)
    want = [lone.compute(b) for b, _ in work]
    import os
    keep_env = os.environ.get("PHMM_ROUTE_SHARED")
    os.environ["PHMM_ROUTE_SHARED"] = "4"                    # (opt-in since round 6: which regions share a flush depends on timing)
    try:
        engines = [HipPairHMMEngine(0) for _ in range(12)]   # the switch is read at phmm_create
    finally:
        if keep_env is not None:
            os.environ["PHMM_ROUTE_SHARED"] = keep_env
        else:
            del os.environ["PHMM_ROUTE_SHARED"]
    errors = []

    def worker(e):
        try:
            for rep in range(4):
                for (b, oracle_lk), w in zip(work, want):
                    got = e.compute(b)
                    assert np.max(np.abs(got - oracle_lk)) <= TOL
                    assert np.max(np.abs(got - w)) <= 1e-11   # (a combined flush may sweep a region with another lane geometry)
        except Exception as ex:
            errors.append(ex)

    try:
        staged0 = [e.stat("staged_bytes") for e in engines]
        th = [threading.Thread(target=worker, args=(e,)) for e in engines]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors[0]
        # none of the twelve staged anything itself: the calls went through the backing handle's lanes
        assert [e.stat("staged_bytes") for e in engines] == staged0
        # a handle whose switches were set keeps its calls to itself (A/B runs, the tests' switches)
        engines[0].set_switch("force_L", 0)
        b, _ = work[0]
        assert np.array_equal(engines[0].compute(b), want[0])
        assert engines[0].stat("staged_bytes") > staged0[0]
        # errors of a routed call come back on the caller's own handle: one that only the FLUSH can raise (the reference's `<= 0`
        # assert, pair_hmm.rs:478-481 -- the arguments are valid, the entry point's own checks pass, the backing handle's lane fails)
        hap = np.full(40, ord("A"), np.uint8)
        weird = RegionBatch.from_regions([([Read(hap[:8].copy(), np.full(8, 93), np.zeros(8, int), np.zeros(8, int), np.full(8, 10))], [hap])])
        staged1 = engines[1].stat("staged_bytes")
        with pytest.raises(PhmmError) as e:
            engines[1].compute(weird)
        assert e.value.code == _lib.PHMM_ERR_POSITIVE_RESULT and "greater than 0.0" in engines[1].last_error()
        assert engines[1].stat("staged_bytes") == staged1       # (it was the backing handle that ran it)
        assert np.max(np.abs(engines[1].compute(work[1][0]) - want[1])) <= 1e-11   # ... and the handle goes on
    finally:
        for e in engines:
            e.close()
        lone.close()


def test_thirty_two_private_handles_are_not_slower_than_sixteen():
    """tools/threads_bench, 16 and 32 C++ threads with a handle each, the library's defaults (46.7 k -> 24.0 k regions/s
    at 32 private handles).  The cause was the waiters: hipStreamSynchronize spins, 32 spinning threads in a container with 16
    cores' worth of CPU are throttled together with the callers that stage -- a one-shot call now waits in 20 us naps once the
    caller holds more handles than it has cores (phmm_host::wait_stream): phmm_compute alone (`own`) 22 -> 45 k at 32 handles,
    the launched region call (`fused`, PHMM_REGION_SERVER=0) 12.5 -> 22 k, and the default region call goes through the region
    server (45 k).  Nothing is routed or combined for any of that."""
    import os
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "threads_bench")
    for mode, server in (("own", None), ("fused", None), ("fused", "0")):
        env = dict(os.environ, TB_MODE=mode, TB_THREADS="16,32", TMPDIR="/tmp")
        env.pop("PHMM_ROUTE_SHARED", None)
        env.pop("PHMM_REGION_SERVER", None)
        if server is not None:
            env["PHMM_REGION_SERVER"] = server
        r = subprocess.run([exe, "1.5"], capture_output=True, text=True, timeout=300, env=env)
        print(r.stdout, r.stderr[-1000:])
        assert r.returncode == 0, r.stdout + r.stderr
        rate = {int(t): float(v) for t, v in re.findall(r"(\d+) threads:\s+(\d+) regions/s", r.stdout)}
        # (the box has 16 cores: at 32 threads the callers themselves are oversubscribed, and one shared handle shows the same
        # few per cent -- the cliff was a factor of two)
        assert rate[32] >= 0.75 * rate[16], rate
