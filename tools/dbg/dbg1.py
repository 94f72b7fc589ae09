import sys, os, traceback
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from lorikeet_amd import region
from lorikeet_amd.engine import HipPairHMMEngine
from project_scenarios import scenario
from test_region_hip import _cfg, _noisy_quals, _oracle_pipeline, _priorities
eng = HipPairHMMEngine(0)
try:
    for seed, low in [(21, False), (22, True)]:
        sc = scenario(seed, n_regions=5, low_complexity=low)
        b = sc[0]
        mapq = _noisy_quals(b, seed)
        cfg = _cfg(pcr=3, dynamic=True)
        pri = _priorities(b, sc[1], sc[3])
        b_, hc, hs, rh, rs, oc = sc
        got = region.region_compute(eng, cfg, b, mapq, hc, hs, rh, rs, oc, hap_priority=pri)
        print("server jobs", eng.stat("server_jobs"), "all pairs", eng.stat("server_all_pairs"), "broken", eng.stat("server_broken"), flush=True)
        out, keep, best = _oracle_pipeline(cfg, b, mapq, sc[3], pri)
        d = np.abs(got.likelihoods - out)
        assert eng.stat("server_jobs") >= 1
        print("max diff", np.nanmax(d), "nan", np.isnan(got.likelihoods).sum(), "keep eq", np.array_equal(got.keep, keep), "best eq", np.array_equal(got.best.allele_index, best), flush=True)
        if not np.array_equal(got.best.allele_index, best):
            print(got.best.allele_index, best)
        if np.nanmax(d) > 1e-9:
            bad = np.flatnonzero(~(d < 1e-9))
            print("bad idx", bad[:20], got.likelihoods[bad[:10]], out[bad[:10]])
except BaseException as e:
    traceback.print_exc()
sys.stdout.flush()
import faulthandler; faulthandler.enable()
print('closing', flush=True)
eng.close()
print('closed', flush=True)
os._exit(0)
