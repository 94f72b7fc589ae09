"""Developer tool: the whole covered path per region, the way the reference runs it -- T host threads, one handle each, per
region: phmm_engine_compute (qualities, PairHMM, normalise, filter) and phmm_realign_reads (best alleles, alignments,
projection onto the reference) with the likelihoods it just got; host buffers, PCIe included.  Aggregate regions/s and
reads/s.  usage: python tools/region_pipeline.py [seconds per point] [regions per call]"""
import ctypes as C
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lorikeet_amd import HipPairHMMEngine, _lib, synthetic  # noqa: E402

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
per_call = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class Job:
    """One call's worth of regions with every array the two calls take."""

    def __init__(self, seed):
        b = self.b = synthetic.config2(per_call, seed=seed)
        n, nh = b.n_reads, b.n_haps
        self.mapq = np.full(n, 60, np.uint8)
        self.ref = np.zeros(b.n_regions, np.int32)
        self.out, self.keep = np.empty(b.n_out, np.float64), np.zeros(n, np.uint8)
        self.pri = np.zeros(nh, np.int32)
        self.rstart = (1000 * (1 + np.arange(b.n_regions))).astype(np.uint64)
        self.hc_off = np.arange(nh + 1, dtype=np.uint32)
        self.hc = ((np.diff(b.hap_off.astype(np.int64)) << 4) | 0).astype(np.uint32)
        self.hs = np.zeros(nh, np.uint32)
        self.oc_off = np.arange(n + 1, dtype=np.uint32)
        self.oc = ((np.diff(b.read_off.astype(np.int64)) << 4) | 0).astype(np.uint32)
        self.out_off = np.arange(n + 1, dtype=np.uint64) * 8
        self.cig, self.n_cig = np.zeros(n * 8, np.uint32), np.zeros(n, np.uint32)
        self.pos, self.status, self.best = np.zeros(n, np.int64), np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.lk, self.conf = np.zeros(n), np.zeros(n)
        self.prm = _lib.SwParameters(10, -15, -30, -5)

    def run(self, eng, cfg):
        b = self.b
        st = eng.lib.phmm_engine_compute(eng._h, C.byref(cfg), b.n_regions, pp(b.region_read_off, _lib.u32p), pp(b.region_hap_off, _lib.u32p),
                                         pp(b.read_off, _lib.u32p), pp(b.read_bases, _lib.u8p), pp(b.base_q, _lib.u8p), None, None,
                                         pp(self.mapq, _lib.u8p), pp(b.hap_off, _lib.u32p), pp(b.hap_bases, _lib.u8p), pp(self.ref, i32p),
                                         pp(b.out_off, _lib.u64p), pp(self.out, _lib.f64p), pp(self.keep, _lib.u8p))
        assert st == 0, eng.last_error()
        st = eng.lib.phmm_realign_reads(eng._h, b.n_regions, pp(b.region_read_off, _lib.u32p), pp(b.region_hap_off, _lib.u32p),
                                        pp(b.read_off, _lib.u32p), pp(b.read_bases, _lib.u8p), pp(b.hap_off, _lib.u32p), pp(b.hap_bases, _lib.u8p),
                                        pp(b.out_off, _lib.u64p), pp(self.out, _lib.f64p), pp(self.keep, _lib.u8p), pp(self.pri, i32p), 0.2,
                                        C.byref(self.prm), 0, pp(self.ref, i32p), pp(self.rstart, _lib.u64p), pp(self.hc_off, _lib.u32p),
                                        pp(self.hc, _lib.u32p), pp(self.hs, _lib.u32p), pp(self.oc_off, _lib.u32p), pp(self.oc, _lib.u32p),
                                        pp(self.out_off, _lib.u64p), pp(self.cig, _lib.u32p), pp(self.n_cig, _lib.u32p), pp(self.pos, i64p),
                                        pp(self.status, i32p), pp(self.best, i32p), pp(self.lk, _lib.f64p), pp(self.conf, _lib.f64p))
        assert st == 0, eng.last_error()


cfg = _lib.EngineConfig()
cfg.constant_gcp, cfg.pcr_error_model, cfg.base_quality_score_threshold = 10, 3, 18
cfg.dynamic_read_disqualification, cfg.symmetrically_normalize_alleles_to_reference = 1, 1
cfg.log10_global_read_mismapping_rate = -4.5 * math.log10(math.e)
cfg.read_disqualification_scale, cfg.expected_error_rate_per_base = 1.0, 0.02
for T in (1, 4, 8, 16, 32):
    engines = [HipPairHMMEngine(0) for _ in range(T)]
    jobs = [[Job(1000 * i + k) for k in range(4)] for i in range(T)]
    for e, js in zip(engines, jobs):
        js[0].run(e, cfg)
    counts = [0] * T
    stop = time.perf_counter() + dur

    def work(i):
        n = 0
        while time.perf_counter() < stop:
            jobs[i][n % 4].run(engines[i], cfg)
            n += 1
        counts[i] = n

    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    n = sum(counts)
    b0 = jobs[0][0].b
    print("%2d threads x %d regions per call: %7.0f regions/s, %6.2f M reads/s, PairHMM part %7.1f GCUPS (%.0f us per call per thread); "
          "realigned %d of %d reads of the last call" % (T, per_call, n * per_call / dt, n * b0.n_reads / dt / 1e6, n * b0.cells() / dt / 1e9,
                                                          dt * T / n * 1e6, int(np.sum(jobs[0][0].status == 0)), b0.n_reads), flush=True)
    for e in engines:
        e.close()
