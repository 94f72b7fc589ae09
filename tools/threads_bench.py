"""Developer tool: the reference's call pattern -- T host threads, one handle each, one region per phmm_compute call
(host buffers, PCIe included).  Aggregate regions/s.  usage: python tools/threads_bench.py [seconds per point]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lorikeet_amd import HipPairHMMEngine, synthetic

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
regions = [synthetic.config2(1, seed=100 + i) for i in range(8)]
cells = regions[0].cells()
for T in (1, 2, 4, 8, 16, 32):
    engines = [HipPairHMMEngine(0) for _ in range(T)]
    for e in engines:
        e.compute(regions[0])
    counts = [0] * T
    stop = time.perf_counter() + dur

    def work(i):
        e, n, k = engines[i], 0, i
        while time.perf_counter() < stop:
            e.compute(regions[k % len(regions)])
            n += 1
            k += 1
        counts[i] = n

    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    n = sum(counts)
    print("%2d threads: %7.0f regions/s  %7.1f GCUPS  (%.1f us per call per thread)" % (T, n / dt, n * cells / dt / 1e9, dt * T / n * 1e6), flush=True)
    for e in engines:
        e.close()
