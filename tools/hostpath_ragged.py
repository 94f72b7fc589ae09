import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from lorikeet_amd import HipPairHMMEngine, synthetic
b = synthetic.ragged()
eng = HipPairHMMEngine(0)
for i in range(3):
    t = time.perf_counter(); out = eng.compute(b); print("call %d: %.2f ms" % (i, (time.perf_counter() - t) * 1e3), file=sys.stderr)
eng.set_switch("trace", 1)
t = time.perf_counter(); out = eng.compute(b); print("traced call: %.2f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
