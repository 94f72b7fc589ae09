"""Developer tool: phmm_engine_compute from host buffers (PCIe included), one-shot vs chunk-pipelined.
usage: python tools/engine_call.py [regions ...]"""
import ctypes as C, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lorikeet_amd import HipPairHMMEngine, _lib, synthetic

eng = HipPairHMMEngine(0)
cfg = _lib.EngineConfig()
cfg.constant_gcp, cfg.pcr_error_model, cfg.base_quality_score_threshold = 10, 3, 18
cfg.dynamic_read_disqualification, cfg.symmetrically_normalize_alleles_to_reference = 1, 1
cfg.log10_global_read_mismapping_rate = -4.5 * math.log10(math.e)
cfg.read_disqualification_scale, cfg.expected_error_rate_per_base = 1.0, 0.02
pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
for n in [int(a) for a in sys.argv[1:]] or [256, 1024]:
    b = synthetic.config2(n, seed=77)
    mapq = np.full(b.n_reads, 60, np.uint8)
    ref = np.zeros(b.n_regions, np.int32)
    out = np.empty(b.n_out, np.float64)
    keep = np.zeros(b.n_reads, np.uint8)
    args = (eng._h, C.byref(cfg), b.n_regions, pp(b.region_read_off, _lib.u32p), pp(b.region_hap_off, _lib.u32p),
            pp(b.read_off, _lib.u32p), pp(b.read_bases, _lib.u8p), pp(b.base_q, _lib.u8p), None, None, pp(mapq, _lib.u8p),
            pp(b.hap_off, _lib.u32p), pp(b.hap_bases, _lib.u8p), pp(ref, C.POINTER(C.c_int32)), pp(b.out_off, _lib.u64p),
            pp(out, _lib.f64p), pp(keep, _lib.u8p))
    for mode in ("pipelined", "one shot"):
        eng.set_switch("no_pipeline", 1 if mode == "one shot" else 0)
        assert eng.lib.phmm_engine_compute(*args) == 0, eng.last_error()
        t = time.perf_counter()
        for _ in range(5):
            assert eng.lib.phmm_engine_compute(*args) == 0
        dt = (time.perf_counter() - t) / 5
        print("%5d regions  %-9s %8.2f ms  %7.1f GCUPS incl. PCIe, kept %.3f" % (n, mode, dt * 1e3, b.cells() / dt / 1e9, keep.mean()), flush=True)
    eng.set_switch("no_pipeline", 0)
