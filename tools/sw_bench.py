"""Developer tool: phmm_sw_align on the read -> haplotype realignment shape (reads of config-2 regions against the first
haplotype of their region; SoftClip, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS), host buffers.  usage:
python tools/sw_bench.py [regions] [strategy] [haps]  (run under rocprofv3 --kernel-trace for the kernel's own time)
`haps`: the haplotype -> reference shape instead (CigarUtils::calculate_cigar, src/reads/cigar_utils.rs:358-405): every
haplotype of config-5 regions (64 x 400 bases) against the first one, NEW_SW_PARAMETERS."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lorikeet_amd import HipPairHMMEngine, _lib, synthetic  # noqa: E402

nreg = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
strategy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
haps_mode = len(sys.argv) > 3 and sys.argv[3] == "haps"
if haps_mode:
    sub = synthetic.config5(nreg, seed=1000)
    n = sub.n_haps
    alt_off, alt = sub.hap_off, sub.hap_bases
    reg = np.repeat(np.arange(sub.n_regions), np.diff(sub.region_hap_off.astype(np.int64)))
    hlen = 400
else:
    sub = synthetic.config2(nreg, seed=1000)
    n = sub.n_reads
    alt_off, alt = sub.read_off, sub.read_bases
    reg = np.repeat(np.arange(sub.n_regions), np.diff(sub.region_read_off.astype(np.int64)))
    hlen = 300
fh = sub.region_hap_off[:-1].astype(np.int64)[reg]
hb = sub.hap_off.astype(np.int64)
ref_off = np.concatenate([[0], np.cumsum(hb[fh + 1] - hb[fh])]).astype(np.uint32)
idx = (hb[fh][:, None] + np.arange(hlen)[None, :]).reshape(-1)   # every haplotype of these sets has the same length
ref = np.ascontiguousarray(sub.hap_bases[idx])
cells = int(np.sum((hb[fh + 1] - hb[fh]) * np.diff(alt_off.astype(np.int64))))
cap = 16
cig_off = np.arange(n + 1, dtype=np.uint64) * cap
cigar, n_cig, off = np.zeros(n * cap, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
prm = _lib.SwParameters(200, -150, -260, -11) if haps_mode else _lib.SwParameters(10, -15, -30, -5)
eng = HipPairHMMEngine(0)
eng.set_switch("sw_clock", 1)
pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
args = (eng._h, n, pp(ref_off, _lib.u32p), pp(ref, _lib.u8p), pp(alt_off, _lib.u32p), pp(alt, _lib.u8p), C.byref(prm), strategy,
        pp(cig_off, _lib.u64p), pp(cigar, _lib.u32p), pp(n_cig, _lib.u32p), pp(off, C.POINTER(C.c_int32)))
assert eng.lib.phmm_sw_align(*args) == 0, eng.last_error()
t = time.perf_counter()
for _ in range(5):
    assert eng.lib.phmm_sw_align(*args) == 0
dt = (time.perf_counter() - t) / 5
print("%d alignments, %.3g cells: %.3f ms per call, %.1f GCUPS-i32, %.2f M alignments/s (host buffers, PCIe included); "
      "CIGAR element counts: %s" % (n, cells, dt * 1e3, cells / dt / 1e9, n / dt / 1e6, np.bincount(n_cig)[:8].tolist()))
kus = eng.stat("sw_kernel_us")
print("kernel (HIP events around the launch): %.3f ms = %.1f GCUPS-i32; backtrack flags written: %.2f GB; shader clock %d MHz" %
      (kus / 1e3, cells / kus / 1e3, eng.stat("sw_backtrack_bytes") / 1e9, eng.stat("sw_clock_mhz")))
import json  # noqa: E402
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402
print(json.dumps({"sw_bench": {"alignments": int(n), "cells": cells, "calls": 6, "ms_per_call": round(dt * 1e3, 3), "kernel_ms": round(kus / 1e3, 3),
                               "backtrack_bytes": int(eng.stat("sw_backtrack_bytes")), "clock_mhz": int(eng.stat("sw_clock_mhz")),
                               "src_hash": bench.source_hash("sw")}}))
