"""Developer tool: 20 resident launches of a 200-region ragged batch (what the first chunk of a host-buffer call looks like)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from lorikeet_amd import HipPairHMMEngine, synthetic  # noqa: E402

eng = HipPairHMMEngine(0)
b = synthetic.ragged(int(sys.argv[1]) if len(sys.argv) > 1 else 200, seed=4442)
plan = eng.plan(b)
dev = torch.device("cuda:0")
t = {k: torch.from_numpy(getattr(b, k)).to(dev) for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
out = torch.empty(b.n_out, dtype=torch.float64, device=dev)
plan.bind_torch(t, out)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(20):
        plan.launch(st.cuda_stream)
        st.synchronize()
plan.status()
plan.close()
eng.close()
