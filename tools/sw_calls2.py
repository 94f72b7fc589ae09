"""Developer tool: what in bench.py's process doubles phmm_sw_align's host time (3.5 -> 7.1 ms in some runs)?  The aligner's calls
timed on a fresh engine, then after each kind of earlier work on the same engine."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lorikeet_amd import HipPairHMMEngine, _lib, synthetic  # noqa: E402

batch = synthetic.config2(1024, seed=1000)
sub = batch
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


def sw(label):
    ts = []
    for i in range(8):
        t = time.perf_counter()
        assert eng.lib.phmm_sw_align(*args) == 0
        ts.append((time.perf_counter() - t) * 1e3)
    print("%-48s %s" % (label, " ".join("%.2f" % x for x in ts)), flush=True)


sw("fresh engine")
for _ in range(3):
    eng.compute(batch)
sw("after phmm_compute (host path, 1 024 regions)")
plan = eng.plan(batch)
plan.upload()
for _ in range(5):
    plan.launch()
plan.download()
sw("with a resident plan alive")
plan.close()
sw("after closing it")
rb = synthetic.ragged()
for _ in range(3):
    eng.compute(rb)
sw("after the mixed batch through host buffers")
import torch  # noqa: E402
x = torch.zeros(1 << 28, dtype=torch.uint8, device="cuda")
y = x.cpu().numpy().sum()
sw("after torch device work + a D2H of 256 MB")
e2 = HipPairHMMEngine(0, f32_first=True)
e2.compute(batch)
sw("with a second engine (f32 first) alive")
eng.set_switch("sw_clock", 1)
sw("switch sw_clock on")
eng.set_switch("sw_lite", 0)
sw("sw_lite 0")
eng.set_switch("sw_lite", -1)
sw("sw_lite -1 again")
