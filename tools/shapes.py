"""Developer tool: throughput of the forward kernels on the BASELINE configs, per forced lanes-per-pair."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lorikeet_amd import HipPairHMMEngine, synthetic

def timed(eng, batch, reps=3):
    reps = max(reps, min(200, int(2e9 / max(batch.cells(), 1))))  # small batches: many launches per measurement
    plan = eng.plan(batch)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    out = torch.empty(batch.n_out, dtype=torch.float64, device=dev)
    plan.bind_torch(t, out)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.launch(st.cuda_stream); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps): plan.launch(st.cuda_stream)
        e1.record(st); st.synchronize()
    if not os.environ.get('SHAPES_NOSTATUS'): plan.status()
    return e0.elapsed_time(e1) / reps, plan.cells, plan.dominant_kernel

cfgs = {"config2x1": lambda: synthetic.config2(1, seed=31), "config2x2": lambda: synthetic.config2(2, seed=32), "config2x4": lambda: synthetic.config2(4, seed=33),
        "config2x8": lambda: synthetic.config2(8, seed=34), "config2x16": lambda: synthetic.config2(16, seed=35),
        "config2x256": lambda: synthetic.config2(256, seed=21), "config2x128": lambda: synthetic.config2(128, seed=22), "config2x64": lambda: synthetic.config2(64, seed=23),
        "config2x1024": lambda: synthetic.config2(1024, seed=1),
        "config3x1024": lambda: synthetic.config3(1024, seed=2),
        "config5x32": lambda: synthetic.config5(32, seed=3),
        "R100_H300x1024": lambda: synthetic.make_regions(1024, 128, 8, 300, 100, 5),
        "R250_H300x512": lambda: synthetic.make_regions(512, 128, 8, 300, 250, 6),
        "Nh2_R150_H300x2048": lambda: synthetic.make_regions(2048, 128, 2, 300, 150, 7),
        "Nh1_R150_H300x4096": lambda: synthetic.make_regions(4096, 128, 1, 300, 150, 11),
        "Nh3_R150_H300x1536": lambda: synthetic.make_regions(1536, 128, 3, 300, 150, 12),
        "Nh5_R150_H300x1024": lambda: synthetic.make_regions(1024, 128, 5, 300, 150, 13),
        "Nh6_R150_H300x1024": lambda: synthetic.make_regions(1024, 128, 6, 300, 150, 14),
        "R150_H600x512": lambda: synthetic.make_regions(512, 64, 8, 600, 150, 9),
        "ragged_small": lambda: synthetic.make_regions(4096, 12, 3, 220, [80, 120, 151], 8)}
only = [a for a in sys.argv[1:] if not a.startswith("--")]
chain_mode = "--chain" in sys.argv  # compare planner default / chained kernel off / forced on instead of forcing L
for name, mk in cfgs.items():
    if only and name not in only: continue
    b = mk()
    if chain_mode:
        for ch in (None, "0", "16", "streams1"):
            os.environ.pop("PHMM_FORCE_STREAMS", None)
            if ch is None: os.environ.pop("PHMM_FORCE_CHAIN", None)
            elif ch == "streams1":  # chained, but one stream per wave (idle haplotype slots)
                os.environ.pop("PHMM_FORCE_CHAIN", None); os.environ["PHMM_FORCE_STREAMS"] = "1"
            else: os.environ["PHMM_FORCE_CHAIN"] = ch
            eng = HipPairHMMEngine(0)
            ms, cells, k = timed(eng, b)
            print("%-20s chain=%-4s %-26s %8.3f ms %8.1f GCUPS" % (name, ch, k, ms, cells / ms / 1e6), flush=True)
            eng.close()
        continue
    for L in (0, 16, 32, 64):
        if L: os.environ["PHMM_FORCE_L"] = str(L)
        else: os.environ.pop("PHMM_FORCE_L", None)
        eng = HipPairHMMEngine(0)
        try:
            ms, cells, k = timed(eng, b)
            print("%-20s L=%-2d %-22s %8.3f ms %8.1f GCUPS" % (name, L, k, ms, cells / ms / 1e6), flush=True)
        except Exception as e:
            print(name, L, "failed:", e)
        eng.close()
