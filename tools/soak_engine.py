"""Developer tool: soak parity of the engine-level call (PCR indel model + quality caps, PairHMM, normalisation,
disqualification decision) against the oracle pipeline, for a wall-clock budget.
usage: python tools/soak_engine.py [seconds] [seed]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lorikeet_amd.likelihood_engine import PairHMMLikelihoodCalculationEngine, PCRErrorModel
import test_engine_hip as T  # the oracle pipeline and the random-region generator of the GPU tests

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t_end = time.time() + budget
n_calls = n_regions = n_reads = n_removed = 0
worst = 0.0
while time.time() < t_end:
    pcr = PCRErrorModel(int(rng.integers(0, 4)))
    dynamic, symmetric, with_tags = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    cfg = dict(gcp=int(rng.choice([10, 10, 8, 20])), cap=-4.5 * math.log10(math.e), pcr=pcr, bq_threshold=int(rng.choice([18, 6, 25])),
               dynamic=dynamic, scale=float(rng.choice([1.0, 0.5, 2.0])), err=0.02, symmetric=symmetric,
               disable_cap=bool(rng.integers(0, 2)))
    eng = PairHMMLikelihoodCalculationEngine(cfg["gcp"], cfg["cap"], pcr, cfg["bq_threshold"], dynamic, cfg["scale"], cfg["err"],
                                             symmetric, cfg["disable_cap"])
    regions = T._random_regions(rng, int(rng.integers(1, 30)), with_tags)
    got = eng.compute_regions(regions)
    want = T._oracle_pipeline(cfg, regions)
    for (gm, gk), (wm, wk) in zip(got, want):
        assert gm.shape == wm.shape
        if gm.size:
            d = float(np.max(np.abs(gm - wm)))
            worst = max(worst, d)
            assert d <= 1e-9, d
        assert np.array_equal(gk, wk)
        n_removed += int((~wk).sum())
        n_reads += len(wk)
    n_calls += 1
    n_regions += len(regions)
print("engine soak ok: %d calls, %d regions, %d reads (%d disqualified), worst |hip - oracle| = %.3g, keep flags identical"
      % (n_calls, n_regions, n_reads, n_removed, worst))
