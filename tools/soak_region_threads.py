"""Developer tool: the region call under concurrency -- T threads with an engine each (up to four: hardware queues of their
own, NOTEBOOK.md 18.6; the in-flight rule then mixes all-pairs calls and chains) and T threads on ONE shared handle
(phmm_region_submit / phmm_wait: combined flushes on the zero-copy path), every result compared field by field with what a
lone caller got for the same region the chain's way.  usage: python tools/soak_region_threads.py [seconds] [threads] [seed]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lorikeet_amd import HipPairHMMEngine, region  # noqa: E402
from project_scenarios import scenario  # noqa: E402
from test_region_hip import _cfg, _equal_calls, _noisy_quals, _priorities  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = _cfg(pcr=2)
jobs = []
lone = HipPairHMMEngine(0)
lone.set_switch("region_sw_all", 0)
for k in range(48):
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig = scenario(seed * 1009 + k, n_regions=1 + k % 3, low_complexity=k % 4 == 0)
    mapq = _noisy_quals(b, k)
    pri = _priorities(b, hap_cigars, ref_hap)
    args = (cfg, b, mapq, hap_cigars, hap_starts, ref_hap, ref_start, orig)
    jobs.append((args, pri, region.region_compute(lone, *args, hap_priority=pri)))
lone.close()


def run(mode):
    engines = [HipPairHMMEngine(0) for _ in range(T)] if mode == "own" else [HipPairHMMEngine(0)] * T
    errs, counts = [], [0] * T
    t_end = time.time() + seconds / 2

    def worker(t):
        try:
            rng = np.random.default_rng(seed * 31 + t)
            while time.time() < t_end:
                args, pri, want = jobs[int(rng.integers(0, len(jobs)))]
                if mode == "own":
                    got = region.region_compute(engines[t], *args, hap_priority=pri)
                else:
                    got = region.region_compute(engines[t], *args, hap_priority=pri, shared=True)
                _equal_calls(got, want)
                counts[t] += 1
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    pairs = sum(e.stat("region_sw_all") for e in set(engines))
    for e in set(engines):
        e.close()
    if errs:
        raise errs[0]
    return sum(counts), pairs


n_own, p_own = run("own")
n_sh, p_sh = run("shared")
print("region-call thread soak ok: %d threads -- an engine each: %d calls (%d with every pair aligned); one shared handle: %d "
      "submissions; all equal field by field to a lone caller's chain" % (T, n_own, p_own, n_sh))
