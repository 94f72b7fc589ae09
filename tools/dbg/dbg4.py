import sys, os, threading
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from lorikeet_amd import region, synthetic
from lorikeet_amd.engine import HipPairHMMEngine
from project_scenarios import scenario
from test_region_hip import _cfg, _noisy_quals
base = HipPairHMMEngine(0)
if sys.argv[1] == "1":
    es = [HipPairHMMEngine(0) for _ in range(8)]
    b = synthetic.make_regions(3, 24, 3, 120, [40, 77], seed=5)
    for e in es:
        e.compute(b)
    print("server jobs", base.stat("server_jobs"), flush=True)
    for e in es:
        e.close()
sc = scenario(900, n_regions=1)
mapq = _noisy_quals(sc[0], 7)
cfg = _cfg(pcr=2)
e2 = [HipPairHMMEngine(0), HipPairHMMEngine(0)]
for e in e2:
    e.set_switch("region_server", 0)
    for _ in range(3):
        region.region_compute(e, cfg, sc[0], mapq, *sc[1:])
print("region_sw_all", [e.stat("region_sw_all") for e in e2], "server jobs", base.stat("server_jobs"), flush=True)
os._exit(0)
