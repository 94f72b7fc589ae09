"""The device-side hand-offs of the small region call (phmm_region.cpp; NOTEBOOK 18.1, 18.7, 19.1) under test on the MI355X:

* phmm_pick_reads waits for the all-pairs aligner on ANOTHER stream through a device counter.  The wait is bounded: out of
  time, the kernel raises a word, ends, and the host runs the call again the chained way.  Forced here by putting the aligner
  BEHIND the waiting kernel on one in-order stream (switch `region_debug_pick` = 1) -- the call must complete, with the
  chain's results, and say so in `region_pick_timeouts`.
* PHMM_MIRROR_CANARY (switch `mirror_canary`): a device store that lands in the pinned mirror after its call has returned,
  or in the staged inputs of a call, fails that call.  The negative control re-creates round 4's bug on purpose (switch
  `region_debug_pick` = 2: the aligner stores two words ~300 us behind counting itself in) and must be caught; without the
  bug thousands of calls of every entry point run clean under the canary.
* tools/threads_bench TB_VERIFY=1 (C++ callers, no interpreter between the calls -- the harness that found that bug) over
  the matrix of tools/run/verify_threads.sh, as a collected test, canary on.

Reference for the sequence these calls replace (strictly ordered there): src/haplotype/haplotype_caller_engine.rs:1311-1357."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from lorikeet_amd import region
from lorikeet_amd.engine import HipPairHMMEngine, PhmmError

from project_scenarios import scenario as _scenario
from test_region_hip import _cfg, _equal_calls, _noisy_quals, _priorities

pytestmark = pytest.mark.gpu
TB = os.path.join(ROOT, "tools", "threads_bench")


def _jobs(seed0, n, regions=lambda k: 1 + (k % 3)):
    jobs = []
    for k in range(n):
        b, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars = _scenario(seed0 + k, n_regions=regions(k))
        jobs.append((b, _noisy_quals(b, seed0 + k), hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars))
    return jobs


def test_pick_reads_gives_up_waiting_and_the_call_completes_the_chained_way():
    cfg = _cfg(pcr=3)
    jobs = _jobs(4100, 6)
    eng = HipPairHMMEngine(0)
    try:
        eng.set_switch("region_sw_all", 0)
        want = [region.region_compute(eng, cfg, *j) for j in jobs]
        eng.set_switch("region_sw_all", 1 << 20)
        eng.set_switch("region_pick_timeout_us", 300)
        eng.set_switch("region_debug_pick", 1)      # the aligner behind the kernel that waits for it, on ONE in-order stream
        n_all, n_out = eng.stat("region_sw_all"), eng.stat("region_pick_timeouts")
        for j, w in zip(jobs, want):
            _equal_calls(region.region_compute(eng, cfg, *j), w)
        assert eng.stat("region_sw_all") == n_all + len(jobs)            # every call tried the all-pairs way ...
        assert eng.stat("region_pick_timeouts") == n_out + len(jobs)     # ... ran out of time, and came back right as the chain
        # and the handle is as good as new afterwards: the counters the two streams share are consistent again
        eng.set_switch("region_debug_pick", 0)
        eng.set_switch("region_pick_timeout_us", 5000)
        for _ in range(3):
            for j, w in zip(jobs, want):
                _equal_calls(region.region_compute(eng, cfg, *j), w)
        assert eng.stat("region_pick_timeouts") == n_out + len(jobs)
        assert eng.stat("region_sw_all") == n_all + 4 * len(jobs)
    finally:
        eng.close()


def test_the_two_streams_on_one_hardware_queue_complete():
    """A child process whose runtime has ONE hardware queue for ordinary streams (GPU_MAX_HW_QUEUES=1) and whose handles stay
    off the device's queue pool: whatever the runtime multiplexes, every call returns (the bounded wait), results equal."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from lorikeet_amd import region
from lorikeet_amd.engine import HipPairHMMEngine
from project_scenarios import scenario
from test_region_hip import _cfg, _equal_calls, _noisy_quals
cfg = _cfg(pcr=3)
eng = HipPairHMMEngine(0)
eng.set_switch("region_pick_timeout_us", 2000)
jobs = []
for k in range(5):
    b, hc, hs, rh, rs, oc = scenario(4200 + k, n_regions=1 + k %% 2)
    jobs.append((b, _noisy_quals(b, k), hc, hs, rh, rs, oc))
eng.set_switch("region_sw_all", 0)
want = [region.region_compute(eng, cfg, *j) for j in jobs]
eng.set_switch("region_sw_all", 1 << 20)
for rep in range(20):
    for j, w in zip(jobs, want):
        _equal_calls(region.region_compute(eng, cfg, *j), w)
print("calls", 20 * len(jobs), "all-pairs", eng.stat("region_sw_all"), "timeouts", eng.stat("region_pick_timeouts"))
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1", PHMM_REGION_OWN_QUEUE="0", PHMM_MIRROR_CANARY="1")
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "calls 100" in r.stdout


def test_the_canary_catches_a_store_that_lands_after_its_call():
    """Negative control: round 4's bug re-created on purpose (two words stored ~300 us behind the count).  Calls alternate between
    a small and a larger region, so the late words of the small call land in the larger one's staged inputs, or in the
    small call's poisoned result block."""
    cfg = _cfg(pcr=3)
    small = _jobs(4300, 1, regions=lambda k: 1)[0]
    large = _jobs(4310, 1, regions=lambda k: 4)[0]
    eng = HipPairHMMEngine(0)
    try:
        eng.set_switch("region_sw_all", 1 << 20)
        eng.set_switch("mirror_canary", 1)
        for _ in range(50):                         # without the bug: clean
            region.region_compute(eng, cfg, *small)
            region.region_compute(eng, cfg, *large)
        eng.set_switch("region_debug_pick", 2)
        caught = None
        for _ in range(200):
            try:
                region.region_compute(eng, cfg, *small)
                region.region_compute(eng, cfg, *large)
            except PhmmError as e:
                caught = str(e)
                break
        assert caught and "PHMM_MIRROR_CANARY" in caught, caught
    finally:
        eng.close()


def test_every_entry_point_is_clean_under_the_canary():
    from lorikeet_amd import synthetic
    cfg = _cfg(pcr=1)
    jobs = _jobs(4400, 8)
    batches = [synthetic.make_regions(1 + k % 3, 8 + 5 * k, 2 + k % 4, 120 + 20 * k, [40, 80, 100], seed=900 + k) for k in range(6)]
    eng = HipPairHMMEngine(0)
    try:
        eng.set_switch("mirror_canary", 1)
        first = [region.region_compute(eng, cfg, *j) for j in jobs]
        lk0 = [eng.compute(b) for b in batches]
        for rep in range(30):
            eng.set_switch("region_sw_all", (0, -1, 1 << 20)[rep % 3])
            for j, w in zip(jobs, first):
                _equal_calls(region.region_compute(eng, cfg, *j), w)
            for b, w in zip(batches, lk0):
                assert np.array_equal(eng.compute(b), w)
    finally:
        eng.close()


# (fused / gshared are the calls with device-side hand-offs between queues: 3 s per point; the others 1 s)
MODES = [("own", 1.0), ("shared", 1.0), ("pipeline", 1.0), ("fused", 3.0), ("gshared", 3.0)]
SHAPES = [("config2", ["128", "8", "150", "300", "1"]), ("small", ["30", "3", "150", "300", "1"]), ("ragged", None),
          ("four_per_call", ["128", "8", "150", "300", "4"])]


@pytest.mark.parametrize("mode,seconds", MODES)
@pytest.mark.parametrize("shape,args", SHAPES)
def test_cpp_callers_verify_every_call_against_its_first_pass(mode, seconds, shape, args):
    """tools/threads_bench TB_VERIFY=1: T C++ threads call one entry point as fast as they can, every call's results are
    held against the same region's first pass (likelihoods 1e-9, everything discrete equal); 1, 2, 3, 4, 8, 16 threads."""
    assert os.path.exists(TB), "run __graft_entry__.build()"
    seconds = float(os.environ.get("PHMM_VERIFY_SECONDS", seconds))
    env = dict(os.environ, TB_VERIFY="1", TB_MODE=mode, TB_THREADS="1,2,3,4,8,16", PHMM_MIRROR_CANARY="1", TMPDIR="/tmp")
    env.pop("PHMM_ROUTE_SHARED", None)      # the library's default: own / fused route through the shared lanes from five threads on
    if args is None:
        env["TB_SHAPE"] = "ragged"
    r = subprocess.run([TB, str(seconds)] + (args or []), capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout, r.stderr[-3000:])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "TB_VERIFY" not in r.stderr and "PHMM_MIRROR_CANARY" not in r.stderr
    assert r.stdout.count("regions/s") == 6
