"""Developer tool: phmm_compute latency for mid-size batches of config-2 regions (one-shot vs pipelined chunks)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lorikeet_amd import HipPairHMMEngine, synthetic

eng = HipPairHMMEngine(0, f32_first=bool(os.environ.get("F32")))  # F32=1: the f32-first mode
for n in [int(a) for a in sys.argv[1:]] or (32, 64, 128, 256, 512, 1024, 4096):
    b = synthetic.config2(n, seed=n)
    for _ in range(3):
        eng.compute(b)
    reps = max(3, 2000 // n)
    t = time.perf_counter()
    for _ in range(reps):
        eng.compute(b)
    dt = (time.perf_counter() - t) / reps
    print("%5d regions %9.1f us/call %7.2f us/region %8.1f GCUPS incl. PCIe" % (n, dt * 1e6, dt * 1e6 / n, b.cells() / dt / 1e9), flush=True)
