"""GPU probe (developer tool): parity spot checks + timing of the forward kernels per forced shape."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lorikeet_amd import HipPairHMMEngine, synthetic
from lorikeet_amd.batch import Read, RegionBatch
from oracle import oracle as O


def kat(eng):
    rows = O.load_kat(os.path.join(ROOT, "tests/golden/pairhmm-testdata.txt"))
    b = RegionBatch.from_regions([([Read(r["read"], r["qual"], r["ins"], r["dele"], r["gcp"])], [r["hap"]]) for r in rows])
    got = eng.compute(b)
    exp = np.array([r["expected"] for r in rows])
    orc = O.compute_batch(b.as_dict())
    return float(np.max(np.abs(got - exp))), float(np.max(np.abs(got - orc)))


def timed(eng, batch, reps=5):
    plan = eng.plan(batch)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    out = torch.empty(batch.n_out, dtype=torch.float64, device=dev)
    plan.bind_torch(t, out)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        assert s != 0
        plan.launch(s); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            plan.launch(s)
        e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / reps
    plan.status()
    return ms, plan.cells, plan.dominant_kernel, out.cpu().numpy()


if __name__ == "__main__":
    res = {}
    for L in (0, 16, 32, 64):
        if L: os.environ["PHMM_FORCE_L"] = str(L)
        else: os.environ.pop("PHMM_FORCE_L", None)
        eng = HipPairHMMEngine(0)
        res["kat_L%d" % L] = kat(eng)
        b = synthetic.make_regions(3, 20, 5, 90, [30, 50, 70], seed=5)
        d = float(np.max(np.abs(eng.compute(b) - O.compute_batch(b.as_dict()))))
        res["rand_L%d" % L] = d
        print("L", L, res["kat_L%d" % L], d, flush=True)
        if L == 0:
            g = eng.compute(b); w = O.compute_batch(b.as_dict())
            print("   sample got", g[:4], "want", w[:4], "nz diffs", int(np.count_nonzero(g - w)), "of", g.size)
            e2 = HipPairHMMEngine(0, do_not_use_tristate_correction=True)
            g2 = e2.compute(b); w2 = O.compute_batch(b.as_dict(), disable_tristate=True)
            print("   no-tristate got", g2[:3], "want", w2[:3], "max d", float(np.max(np.abs(g2 - w2))))
        nreg = int(os.environ.get("PROBE_REGIONS", "512"))
        big = synthetic.config2(nreg, seed=1)
        ms, cells, name, out = timed(eng, big)
        res["time_L%d" % L] = dict(ms=ms, gcups=cells / ms / 1e6, kernel=name)
        print("  batch", nreg, "regions:", name, "%.3f ms" % ms, "%.1f GCUPS" % (cells / ms / 1e6), flush=True)
        one = synthetic.config2(1, seed=1)
        ms, cells, name, out1 = timed(eng, one, reps=20)
        res["single_L%d" % L] = dict(us=ms * 1e3, gcups=cells / ms / 1e6, kernel=name)
        print("  single region:", name, "%.1f us" % (ms * 1e3), "%.1f GCUPS" % (cells / ms / 1e6), flush=True)
        if L == 0:
            want = O.compute_batch(one.as_dict(), n_threads=8)
            print("  single-region parity vs oracle:", float(np.max(np.abs(out1 - want))))
        eng.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out/probe.json"), "w"), indent=1)
