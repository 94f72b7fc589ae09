"""Developer tool: one phmm_compute call over the ragged mix through host buffers, 25 calls, on the library in place."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from lorikeet_amd import HipPairHMMEngine, synthetic  # noqa: E402

eng = HipPairHMMEngine(0)
rag = synthetic.ragged()
for _ in range(3):
    eng.compute(rag)
t = []
for _ in range(25):
    t0 = time.perf_counter()
    eng.compute(rag)
    t.append((time.perf_counter() - t0) * 1e3)
t = np.sort(t)
print("ragged host call: best %.3f  p25 %.3f  median %.3f  p75 %.3f ms" % (t[0], t[6], t[12], t[18]))
