import sys, os, traceback
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from lorikeet_amd import region
from lorikeet_amd.engine import HipPairHMMEngine
from project_scenarios import scenario
from test_region_hip import _cfg, _noisy_quals, _priorities
eng = HipPairHMMEngine(0)
seed = int(sys.argv[1]); n = int(sys.argv[2])
sc = scenario(seed, n_regions=n, low_complexity=False)
b = sc[0]
mapq = _noisy_quals(b, seed)
cfg = _cfg(pcr=3, dynamic=True)
pri = _priorities(b, sc[1], sc[3])
b_, hc, hs, rh, rs, oc = sc
try:
    for i in range(3):
        got = region.region_compute(eng, cfg, b, mapq, hc, hs, rh, rs, oc, hap_priority=pri)
    print("ok jobs", eng.stat("server_jobs"), "all pairs", eng.stat("server_all_pairs"), flush=True)
except BaseException as e:
    print("FAIL", repr(e), flush=True)
os._exit(0)
