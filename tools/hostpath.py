"""Developer tool: latency / throughput of the synchronous host-buffer path (phmm_compute)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lorikeet_amd import HipPairHMMEngine, synthetic

eng = HipPairHMMEngine(0)
for name, b, reps in (("1 region 128x8", synthetic.config2(1, seed=1), 200),
                      ("3 reads x 2 haps", synthetic.make_regions(1, 3, 2, 300, 150, seed=2), 200),
                      ("64 regions", synthetic.config2(64, seed=3), 20),
                      ("1024 regions", synthetic.config2(1024, seed=4), 5)):
    eng.compute(b)
    t = time.perf_counter()
    for _ in range(reps):
        eng.compute(b)
    dt = (time.perf_counter() - t) / reps
    print("%-18s %9.1f us/call  %8.1f GCUPS (PCIe + plan + alloc inclusive)" % (name, dt * 1e6, b.cells() / dt / 1e9), flush=True)
