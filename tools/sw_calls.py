"""Developer tool: per-call times of phmm_sw_align on the read -> haplotype shape (131 072 alignments), 40 calls in a row --
is the host pipeline's time stable from call to call?  usage: python tools/sw_calls.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lorikeet_amd import HipPairHMMEngine, _lib, synthetic  # noqa: E402

sub = synthetic.config2(1024, seed=1000)
n = sub.n_reads
reg = np.repeat(np.arange(sub.n_regions), np.diff(sub.region_read_off.astype(np.int64)))
fh = sub.region_hap_off[:-1].astype(np.int64)[reg]
hb = sub.hap_off.astype(np.int64)
ref_off = np.concatenate([[0], np.cumsum(hb[fh + 1] - hb[fh])]).astype(np.uint32)
ref = np.ascontiguousarray(sub.hap_bases[(hb[fh][:, None] + np.arange(300)[None, :]).reshape(-1)])
cig_off = np.arange(n + 1, dtype=np.uint64) * 16
cigar, n_cig, off = np.zeros(n * 16, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
prm = _lib.SwParameters(10, -15, -30, -5)
eng = HipPairHMMEngine(0)
pp = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
args = (eng._h, n, pp(ref_off, _lib.u32p), pp(ref, _lib.u8p), pp(sub.read_off, _lib.u32p), pp(sub.read_bases, _lib.u8p), C.byref(prm), 0,
        pp(cig_off, _lib.u64p), pp(cigar, _lib.u32p), pp(n_cig, _lib.u32p), pp(off, C.POINTER(C.c_int32)))
ts = []
for i in range(40):
    t = time.perf_counter()
    assert eng.lib.phmm_sw_align(*args) == 0
    ts.append((time.perf_counter() - t) * 1e3)
print("per call (ms):", " ".join("%.2f" % x for x in ts))
print("kernel us of the last call:", eng.stat("sw_kernel_us"))
if len(sys.argv) > 1:   # the same with a big resident allocation and a second engine alive, like bench.py's process
    import torch
    x = torch.empty(int(sys.argv[1]) << 20, dtype=torch.uint8, device="cuda")
    ts = []
    for i in range(20):
        t = time.perf_counter()
        assert eng.lib.phmm_sw_align(*args) == 0
        ts.append((time.perf_counter() - t) * 1e3)
    print("with torch + %s MB resident:" % sys.argv[1], " ".join("%.2f" % x for x in ts))
