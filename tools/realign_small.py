"""Developer tool: phmm_realign_to_best one region per call (the reference's call pattern: a region per rayon task), by
region size -- latency of the whole call (best alleles + alignments, host buffers)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lorikeet_amd import HipPairHMMEngine, _lib, synthetic  # noqa: E402

eng = HipPairHMMEngine(0)
eng.set_switch("sw_clock", 1)
i32p = C.POINTER(C.c_int32)
pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(1, 16, 2), (1, 128, 8), (1, 1024, 8), (8, 128, 8), (64, 128, 8)]
for nreg, nr, nh in SHAPES:
    b = synthetic.make_regions(nreg, nr, nh, 300, 150, seed=5)
    lk = eng.compute(b)
    n = b.n_reads
    cap = 16
    cig_off = np.arange(n + 1, dtype=np.uint64) * cap
    cigar, n_cig, off = np.zeros(n * cap, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
    bi, bl, bc = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
    pri = np.zeros(b.n_haps, np.int32)
    prm = _lib.SwParameters(10, -15, -30, -5)
    args = (eng._h, b.n_regions, pp(b.region_read_off, _lib.u32p), pp(b.region_hap_off, _lib.u32p), pp(b.read_off, _lib.u32p),
            pp(b.read_bases, _lib.u8p), pp(b.hap_off, _lib.u32p), pp(b.hap_bases, _lib.u8p), pp(b.out_off, _lib.u64p), pp(lk, _lib.f64p), None,
            pp(pri, i32p), 0.2, C.byref(prm), 0, pp(cig_off, _lib.u64p), pp(cigar, _lib.u32p), pp(n_cig, _lib.u32p), pp(off, i32p),
            pp(bi, i32p), pp(bl, _lib.f64p), pp(bc, _lib.f64p))
    for _ in range(5):
        assert eng.lib.phmm_realign_to_best(*args) == 0, eng.last_error()
    reps = 200
    t = time.perf_counter()
    for _ in range(reps):
        eng.lib.phmm_realign_to_best(*args)
    dt = (time.perf_counter() - t) / reps
    print("%3d regions x %4d reads x %d haps: %8.1f us per call, %7.2f M reads/s, kernels %d us at %d MHz" %
          (nreg, nr, nh, dt * 1e6, n / dt / 1e6, eng.stat("sw_kernel_us"), eng.stat("sw_clock_mhz")))
