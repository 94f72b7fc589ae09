"""Developer tool: the ragged mix (resident launches and one phmm_compute call over host buffers) and the headline batch on the
library that is in place -- run once per variant by tools/ab/ragged_ab.sh, all on ONE box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lorikeet_amd import HipPairHMMEngine, synthetic  # noqa: E402


def resident(eng, batch, reps):
    plan = eng.plan(batch)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    out = torch.empty(batch.n_out, dtype=torch.float64, device=dev)
    plan.bind_torch(t, out)
    st = torch.cuda.Stream()
    best = 1e9
    with torch.cuda.stream(st):
        for _ in range(3):
            plan.launch(st.cuda_stream)
        st.synchronize()
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                plan.launch(st.cuda_stream)
            e1.record(st)
            st.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
    plan.status()
    return best, plan.cells, plan.executed_cells if hasattr(plan, "executed_cells") else 0


eng = HipPairHMMEngine(0)
rag = synthetic.ragged()
ms, cells, _ = resident(eng, rag, 10)
print("ragged resident   %7.3f ms  %7.1f GCUPS" % (ms, cells / ms / 1e6))
for _ in range(2):
    eng.compute(rag)
host = []
for _ in range(7):
    t0 = time.perf_counter()
    eng.compute(rag)
    host.append((time.perf_counter() - t0) * 1e3)
print("ragged host call  %7.3f ms  %7.1f GCUPS (best of 7; median %.3f)" % (min(host), cells / min(host) / 1e6, float(np.median(host))))
ms, cells, _ = resident(eng, synthetic.config2(1024, seed=1000), 10)
print("config2 x 1024    %7.3f ms  %7.1f GCUPS" % (ms, cells / ms / 1e6))
one = synthetic.make_regions(2048, 128, 1, 300, [60, 100, 150, 250], 11)
ms, cells, _ = resident(eng, one, 5)
print("1 hap, mixed R    %7.3f ms  %7.1f GCUPS" % (ms, cells / ms / 1e6))
for nreg in (200, 400):
    sub = synthetic.ragged(nreg, seed=4242 + nreg)
    ms, cells, _ = resident(eng, sub, 20)
    print("ragged x %4d      %7.3f ms  %7.1f GCUPS" % (nreg, ms, cells / ms / 1e6))
