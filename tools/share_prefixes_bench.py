import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lorikeet_amd import HipPairHMMEngine, synthetic
dev = torch.device("cuda:0")
eng = HipPairHMMEngine(0)
def run(batch, share, steps=5):
    plan = eng.plan(batch)
    ex = plan.share_prefixes() if share else plan.cells
    tens = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    out = torch.full((batch.n_out,), float("nan"), dtype=torch.float64, device=dev)
    plan.bind_torch(tens, out)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        plan.launch(st.cuda_stream); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(steps): plan.launch(st.cuda_stream)
        e1.record(st); st.synchronize()
    plan.status()
    ms = e0.elapsed_time(e1) / steps
    res = out.cpu().numpy()
    info = (plan.cells, ex, plan.num_launches, ms)
    plan.close()
    return res, info
for name, b in [("config2 x256", synthetic.make_regions(256, 128, 8, 300, [150], seed=1000)),
                ("config2 x1024", synthetic.make_regions(1024, 128, 8, 300, [150], seed=1000)),
                ("config5 x32", synthetic.config("config5", only=(0, 32))),
                ("ragged", synthetic.ragged(1536))]:
    a, ia = run(b, False)
    s, is_ = run(b, True)
    same = np.array_equal(a, s)
    print("%-14s plain %.3f ms (%d launches)  shared %.3f ms (%d launches)  executed %.3f of cells  effective x%.3f  bit-identical %s  max|d| %.3g" % (
        name, ia[3], ia[2], is_[3], is_[2], is_[1] / is_[0], ia[3] / is_[3], same, float(np.nanmax(np.abs(a - s)))))
    assert np.all(np.isfinite(s)) or not np.all(np.isfinite(a))
