"""One call per region for the whole arithmetic path (phmm_region_compute / phmm_region_submit, include/phmm.h):
what the reference does between PairHMMLikelihoodCalculationEngine::compute_read_likelihoods
(src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs:195-242) and
AssemblyBasedCallerUtils::realign_reads_to_their_best_haplotype (src/assembly/assembly_based_caller_utils.rs:208-246),
which follow each other in HaplotypeCallerEngine::call_region (src/haplotype/haplotype_caller_engine.rs:1311-1357).
Everything runs on the MI355X in one enqueue; this file only moves pointers."""
import ctypes as C

import numpy as np

from . import _lib
from .engine import PhmmError
from .realign import LOG_10_INFORMATIVE_THRESHOLD, BestAlleles, ProjectedReads
from .smith_waterman import ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, OverhangStrategy

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _flat(lists, dtype, off_dtype):
    off = np.concatenate([[0], np.cumsum([len(c) for c in lists])]).astype(off_dtype)
    flat = np.concatenate([np.zeros(0, dtype)] + [np.asarray(c, dtype) for c in lists]).astype(dtype)
    return off, flat


class RegionResult:
    """likelihoods: normalised [read][hap] matrices at batch.out_off; keep: evidence flags; best: BestAlleles;
    reads: ProjectedReads (status / new position / new CIGAR per read)."""

    def __init__(self, likelihoods, keep, best, reads):
        self.likelihoods, self.keep, self.best, self.reads = likelihoods, keep, best, reads


def realign_config(parameters=ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, overhang_strategy=OverhangStrategy.SoftClip,
                   threshold=LOG_10_INFORMATIVE_THRESHOLD, skip_single_allele=False):
    rc = _lib.RealignConfig()
    rc.sw_parameters = parameters.as_struct()
    rc.overhang_strategy = OverhangStrategy.NAMES[overhang_strategy] if isinstance(overhang_strategy, str) else int(overhang_strategy)
    rc.flags = _lib.PHMM_REGION_SKIP_SINGLE_ALLELE if skip_single_allele else 0
    rc.informative_threshold = float(threshold)
    return rc


def region_compute(engine, cfg, batch, mapq, hap_cigars, hap_start_wrt_ref, region_ref_hap, region_reference_start, original_cigars,
                   hap_priority=None, read_soft_clip=None, use_indel_quals=True, rcfg=None, capacity=16, shared=False, engines=None):
    """cfg: _lib.EngineConfig; batch: RegionBatch whose base_q / ins_q / del_q are the ORIGINAL qualities (as
    phmm_engine_compute takes them; use_indel_quals=False passes NULL = flat Q45).  shared=True goes through
    phmm_region_submit / phmm_wait on a shared handle; engines=[...] spreads the call over several engines (one per
    device: phmm_region_compute_multi), `engine` being the first of them."""
    n, nh = batch.n_reads, batch.n_haps
    assert len(hap_cigars) == nh and len(original_cigars) == n
    rcfg = rcfg if rcfg is not None else realign_config()
    hc_off, hc = _flat(hap_cigars, np.uint32, np.uint32)
    oc_off, oc = _flat(original_cigars, np.uint32, np.uint32)
    mq = np.ascontiguousarray(mapq, np.uint8)
    hs = np.ascontiguousarray(hap_start_wrt_ref, np.uint32)
    rrh = np.ascontiguousarray(region_ref_hap, np.int32)
    rs = np.ascontiguousarray(region_reference_start, np.uint64)
    pri = None if hap_priority is None else np.ascontiguousarray(hap_priority, np.int32)
    clip = None if read_soft_clip is None else np.ascontiguousarray(read_soft_clip, np.uint32).reshape(-1)
    out, keep = np.full(batch.n_out, np.nan), np.zeros(n, np.uint8)
    best, lk, conf = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
    cap = np.full(n, int(capacity), np.int64)
    for _attempt in range(2):
        out_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
        oc_out, n_out, pos, status = np.zeros(int(out_off[-1]), np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64), np.zeros(n, np.int32)
        args = (engine._h, C.byref(cfg), C.byref(rcfg), batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p),
                _p(batch.read_off, _lib.u32p), _p(batch.read_bases, _lib.u8p), _p(batch.base_q, _lib.u8p),
                _p(batch.ins_q if use_indel_quals else None, _lib.u8p), _p(batch.del_q if use_indel_quals else None, _lib.u8p), _p(mq, _lib.u8p),
                _p(clip, _lib.u32p), _p(batch.hap_off, _lib.u32p), _p(batch.hap_bases, _lib.u8p), _p(rrh, _i32p), _p(batch.out_off, _lib.u64p),
                _p(pri, _i32p), _p(rs, _lib.u64p), _p(hc_off, _lib.u32p), _p(hc, _lib.u32p), _p(hs, _lib.u32p), _p(oc_off, _lib.u32p),
                _p(oc, _lib.u32p), _p(out_off, _lib.u64p), _p(out, _lib.f64p), _p(keep, _lib.u8p), _p(best, _i32p), _p(lk, _lib.f64p),
                _p(conf, _lib.f64p), _p(oc_out, _lib.u32p), _p(n_out, _lib.u32p), _p(pos, _i64p), _p(status, _i32p))
        if engines is not None:
            hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
            code = engine.lib.phmm_region_compute_multi(hs, len(engines), *args[1:])
        elif shared:
            ticket = C.c_uint64(0)
            code = engine.lib.phmm_region_submit(*args, C.byref(ticket))
            if code == _lib.PHMM_OK:
                code = engine.lib.phmm_wait(engine._h, ticket.value)
        else:
            code = engine.lib.phmm_region_compute(*args)
        if code == _lib.PHMM_ERR_CIGAR_CAPACITY:
            cap = np.maximum(cap, n_out.astype(np.int64))
            continue
        if code != _lib.PHMM_OK:
            raise PhmmError(code, engine.last_error())
        cigars = [oc_out[int(out_off[r]):int(out_off[r]) + int(n_out[r])] for r in range(n)]
        return RegionResult(out, keep.astype(bool), BestAlleles(best, lk, conf), ProjectedReads(status, pos, cigars))
    raise PhmmError(_lib.PHMM_ERR_CIGAR_CAPACITY, engine.last_error())
