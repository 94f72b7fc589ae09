"""Developer tool: phmm_compute latency for small batches of config-2 regions (what one phmm_submit flush costs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lorikeet_amd import HipPairHMMEngine, synthetic

eng = HipPairHMMEngine(0)
for n in [int(a) for a in sys.argv[1:]] or (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    b = synthetic.config2(n, seed=n)
    p = eng.plan(b); kern = p.dominant_kernel; nl = p.num_launches; p.close()
    for _ in range(5):
        eng.compute(b)
    reps = 200
    t = time.perf_counter()
    for _ in range(reps):
        eng.compute(b)
    dt = (time.perf_counter() - t) / reps
    print("%3d regions %8.1f us/call %7.1f us/region  %s x%d" % (n, dt * 1e6, dt * 1e6 / n, kern, nl), flush=True)
b = synthetic.config2(int(os.environ.get('TRACE_N', '3')), seed=3)
eng.set_switch("trace", 1)
for _ in range(3):
    eng.compute(b)
