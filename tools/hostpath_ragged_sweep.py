"""Developer tool: the long-tailed mix through host buffers (phmm_compute, 1 536 regions) under the chunk schedule given in
the environment (PHMM_MIXED_FIRST_CHUNK_KB / PHMM_MIXED_CHUNK_KB are read when the library is loaded): best of 6 calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lorikeet_amd import HipPairHMMEngine, synthetic
b = synthetic.ragged()
eng = HipPairHMMEngine(0)
ts = []
for i in range(7):
    t = time.perf_counter(); eng.compute(b); ts.append((time.perf_counter() - t) * 1e3)
print("first %s MB, cap %s MB: best %.2f ms, median %.2f ms = %.0f GCUPS" % (
    os.environ.get("PHMM_MIXED_FIRST_CHUNK_KB", "4096"), os.environ.get("PHMM_MIXED_CHUNK_KB", "32768"),
    min(ts[1:]), sorted(ts[1:])[3], b.cells() / min(ts[1:]) / 1e6))
