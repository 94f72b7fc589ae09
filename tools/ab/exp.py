"""Developer tool: resident ragged batches of three sizes and the host-buffer call on the library in place (one line)."""
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
                st.synchronize()   # (one launch at a time: what a chunk of a host-buffer call sees)
            e1.record(st)
            st.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
    plan.status()
    plan.close()
    return best


eng = HipPairHMMEngine(0)
out = []
for n in (200, 400):
    out.append("x%d %.3f" % (n, resident(eng, synthetic.ragged(n, seed=4242 + n), 20)))
rag = synthetic.ragged()
out.append("full %.3f" % resident(eng, rag, 8))
for _ in range(3):
    eng.compute(rag)
t = []
for _ in range(21):
    t0 = time.perf_counter()
    eng.compute(rag)
    t.append((time.perf_counter() - t0) * 1e3)
out.append("host median %.3f" % float(np.median(t)))
print("  ".join(out))
