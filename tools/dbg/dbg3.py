import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from lorikeet_amd import region
from lorikeet_amd.engine import HipPairHMMEngine
from project_scenarios import scenario
from test_region_hip import _cfg, _noisy_quals, _priorities
eng = HipPairHMMEngine(0)
ref = HipPairHMMEngine(0); ref.set_switch("region_server", 0)
cases = []
for seed, pcr, sym, dyn, low, n in [(1, 3, True, False, False, 1), (2, 0, False, True, False, 3), (3, 1, True, True, True, 1), (4, 2, False, False, True, 5), (5, 3, True, False, False, 7)]:
    sc = scenario(seed, n_regions=n, low_complexity=low)
    b = sc[0]
    mapq = _noisy_quals(b, seed)
    cfg = _cfg(pcr=pcr, symmetric=sym, dynamic=dyn)
    pri = _priorities(b, sc[1], sc[3])
    want = region.region_compute(ref, cfg, *sc[:1], mapq, *sc[1:], hap_priority=pri)
    cases.append((sc, mapq, cfg, pri, want))
bad = 0
t0 = time.time()
it = 0
while time.time() - t0 < float(sys.argv[1]):
    for ci, (sc, mapq, cfg, pri, want) in enumerate(cases):
        it += 1
        if it % 3 == 0:
            time.sleep(0.002)
        got = region.region_compute(eng, cfg, sc[0], mapq, *sc[1:], hap_priority=pri)
        d = np.abs(got.likelihoods - want.likelihoods)
        if not (np.max(d) < 1e-11) or not np.array_equal(got.best.allele_index, want.best.allele_index) or not np.array_equal(got.reads.new_pos, want.reads.new_pos):
            bad += 1
            idx = np.flatnonzero(~(d < 1e-11))
            b = sc[0]
            print("MISMATCH case", ci, "iter", it, "n bad", len(idx), "of", len(d), "first", idx[:12], "got", got.likelihoods[idx[:6]], "want", want.likelihoods[idx[:6]],
                  "out_off", b.out_off, "nr", b.n_reads, "launches", eng.stat("server_launches"), "jobs", eng.stat("server_jobs"), flush=True)
print("iterations", it, "mismatches", bad, "launches", eng.stat("server_launches"), "jobs", eng.stat("server_jobs"), flush=True)
os._exit(0)
