"""Developer tool: a few 1024-region phmm_compute calls, to be run under
`rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` for a timeline of copies and kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lorikeet_amd import HipPairHMMEngine, synthetic

eng = HipPairHMMEngine(0)
b = synthetic.config2(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, seed=4)
for _ in range(4):
    eng.compute(b)
