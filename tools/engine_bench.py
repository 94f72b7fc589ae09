"""Developer tool: time the engine-level call (pre-step + forward + post-step, host buffers) on config-2 regions."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lorikeet_amd import synthetic
from lorikeet_amd.likelihood_engine import PairHMMLikelihoodCalculationEngine, PCRErrorModel
from lorikeet_amd.pair_hmm import Haplotype, HmmRead

n_regions = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = synthetic.config2(n_regions, seed=77)
regions = []
for g in range(b.n_regions):
    reads = []
    for r in range(int(b.region_read_off[g]), int(b.region_read_off[g + 1])):
        s, e = int(b.read_off[r]), int(b.read_off[r + 1])
        reads.append(HmmRead(b.read_bases[s:e].tobytes(), b.base_q[s:e], mapq=60))
    haps = [Haplotype(b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])].tobytes(), a == int(b.region_hap_off[g]))
            for a in range(int(b.region_hap_off[g]), int(b.region_hap_off[g + 1]))]
    regions.append((reads, haps))
eng = PairHMMLikelihoodCalculationEngine(10, -4.5 * math.log10(math.e), PCRErrorModel(int(os.environ.get('EB_PCR', '3'))), 18,
                                         bool(int(os.environ.get('EB_DYN', '1'))), 1.0, 0.02, True, False)
eng.compute_regions(regions[:2])
for rep in range(3):
    t = time.perf_counter()
    res = eng.compute_regions(regions)
    dt = time.perf_counter() - t
    print("engine call: %d regions %.1f ms (python marshalling included), kept %.3f of reads" %
          (n_regions, dt * 1e3, np.mean([k.mean() for _, k in res])), flush=True)
