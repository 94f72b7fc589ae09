"""Host-side mirror of the arithmetic of the reference's `realign_reads_to_their_best_haplotype`
(src/assembly/assembly_based_caller_utils.rs:208-246) over the C ABI: the best allele of every read with ties broken by
priority (AlleleLikelihoods::best_alleles_breaking_ties_main, src/model/allele_likelihoods.rs:457-554, :1043-1095) and
the read's Smith-Waterman alignment to that haplotype (AlignmentUtils::create_read_aligned_to_ref,
src/reads/alignment_utils.rs:40-70).  Everything runs on the MI355X; projecting the read -> haplotype CIGAR onto the
reference (alignment_utils.rs:83-140) is the caller's."""
import ctypes as C

import numpy as np

from . import _lib
from .engine import PhmmError
from .smith_waterman import ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, OverhangStrategy, SmithWatermanAlignmentResult

LOG_10_INFORMATIVE_THRESHOLD = 0.2  # allele_likelihoods.rs:17
_i32p = C.POINTER(C.c_int32)


def haplotype_alignment_tiebreaking_priority(is_ref, n_cigar_elements):
    """assembly_based_caller_utils.rs:187-195: reference term + (1 - CIGAR elements of the haplotype)."""
    return (np.asarray(is_ref, np.int32) != 0).astype(np.int32) + 1 - np.asarray(n_cigar_elements, np.int32)


def reference_tiebreaking_priority(is_ref):
    """assembly_based_caller_utils.rs:197-199."""
    return (np.asarray(is_ref, np.int32) != 0).astype(np.int32)


class BestAlleles:
    """One BestAllele (allele_likelihoods.rs:1119-1166) per read: allele_index (-1 = None), likelihood, confidence."""

    def __init__(self, allele_index, likelihood, confidence):
        self.allele_index, self.likelihood, self.confidence = allele_index, likelihood, confidence

    def is_informative(self):
        return self.confidence > LOG_10_INFORMATIVE_THRESHOLD


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def best_alleles_breaking_ties(engine, batch, likelihoods, hap_priority=None, keep=None, threshold=LOG_10_INFORMATIVE_THRESHOLD):
    """`likelihoods`: the per-region [read][hap] matrices at batch.out_off, as HipPairHMMEngine.compute / the
    engine-level call return them; `keep`: the evidence flags of the engine-level call (or None)."""
    lk = np.ascontiguousarray(likelihoods, dtype=np.float64)
    pri = None if hap_priority is None else np.ascontiguousarray(hap_priority, dtype=np.int32)
    kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    n = batch.n_reads
    best, out_lk, conf = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
    code = engine.lib.phmm_best_alleles(engine._h, batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p),
                                        _p(batch.out_off, _lib.u64p), _p(lk, _lib.f64p), _p(kp, _lib.u8p), _p(pri, _i32p), float(threshold),
                                        _p(best, _i32p), _p(out_lk, _lib.f64p), _p(conf, _lib.f64p))
    if code != _lib.PHMM_OK:
        raise PhmmError(code, engine.last_error())
    return BestAlleles(best, out_lk, conf)


def realign_reads_to_their_best_haplotype(engine, batch, likelihoods, hap_priority=None, keep=None,
                                          parameters=ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS,
                                          overhang_strategy=OverhangStrategy.SoftClip, threshold=LOG_10_INFORMATIVE_THRESHOLD, capacity=16):
    """-> (BestAlleles, [SmithWatermanAlignmentResult or None per read]).  batch.read_bases are the reads without their
    soft clips.  One call: the best alleles index the haplotypes on the device."""
    lk = np.ascontiguousarray(likelihoods, dtype=np.float64)
    pri = None if hap_priority is None else np.ascontiguousarray(hap_priority, dtype=np.int32)
    kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    st = OverhangStrategy.NAMES[overhang_strategy] if isinstance(overhang_strategy, str) else int(overhang_strategy)
    n = batch.n_reads
    best, out_lk, conf = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
    cap = np.full(n, int(capacity), np.int64)
    prm = parameters.as_struct()
    for _attempt in range(2):
        cig_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
        cigar, n_cig, off = np.zeros(int(cig_off[-1]), np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
        code = engine.lib.phmm_realign_to_best(
            engine._h, batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p),
            _p(batch.read_off, _lib.u32p), _p(batch.read_bases, _lib.u8p), _p(batch.hap_off, _lib.u32p), _p(batch.hap_bases, _lib.u8p),
            _p(batch.out_off, _lib.u64p), _p(lk, _lib.f64p), _p(kp, _lib.u8p), _p(pri, _i32p), float(threshold), C.byref(prm), st,
            _p(cig_off, _lib.u64p), _p(cigar, _lib.u32p), _p(n_cig, _lib.u32p), _p(off, _i32p), _p(best, _i32p), _p(out_lk, _lib.f64p),
            _p(conf, _lib.f64p))
        if code == _lib.PHMM_ERR_CIGAR_CAPACITY:
            cap = np.maximum(cap, n_cig.astype(np.int64))
            continue
        if code != _lib.PHMM_OK:
            raise PhmmError(code, engine.last_error())
        res = [SmithWatermanAlignmentResult(cigar[int(cig_off[a]):int(cig_off[a]) + int(n_cig[a])], off[a]) if best[a] >= 0 else None
               for a in range(n)]
        return BestAlleles(best, out_lk, conf), res
    raise PhmmError(_lib.PHMM_ERR_CIGAR_CAPACITY, engine.last_error())


class ProjectedReads:
    """Per read: status (0 realigned, 1 unchanged, < 0 where the reference panics), new position, new CIGAR elements."""

    def __init__(self, status, new_pos, cigars):
        self.status, self.new_pos, self.cigars = status, new_pos, cigars


def project_to_reference(engine, batch, best_allele, alignments, hap_cigars, hap_start_wrt_ref, region_ref_hap, region_reference_start,
                         original_cigars, capacity=None):
    """The rest of AlignmentUtils::create_read_aligned_to_ref (src/reads/alignment_utils.rs:83-165) for every read of the batch
    (phmm_project_to_reference).  `alignments`: the SmithWatermanAlignmentResult per read realign_reads_to_their_best_haplotype
    returned (None where there is none); `hap_cigars` / `original_cigars`: one array of BAM-encoded elements per haplotype / read."""
    n, nh = batch.n_reads, batch.n_haps
    best = np.ascontiguousarray(best_allele, dtype=np.int32)
    sw_n = np.array([0 if a is None else len(a.elements) for a in alignments], np.uint32)
    sw_off = np.concatenate([[0], np.cumsum(sw_n)]).astype(np.uint64)
    sw = np.concatenate([np.zeros(0, np.uint32)] + [a.elements for a in alignments if a is not None]).astype(np.uint32)
    sw_offset = np.array([-1 if a is None else a.alignment_offset for a in alignments], np.int32)
    hc_off = np.concatenate([[0], np.cumsum([len(c) for c in hap_cigars])]).astype(np.uint32)
    hc = np.concatenate([np.zeros(0, np.uint32)] + [np.asarray(c, np.uint32) for c in hap_cigars]).astype(np.uint32)
    oc_off = np.concatenate([[0], np.cumsum([len(c) for c in original_cigars])]).astype(np.uint32)
    oc = np.concatenate([np.zeros(0, np.uint32)] + [np.asarray(c, np.uint32) for c in original_cigars]).astype(np.uint32)
    hs = np.ascontiguousarray(hap_start_wrt_ref, dtype=np.uint32)
    rrh = np.ascontiguousarray(region_ref_hap, dtype=np.int32)
    rs = np.ascontiguousarray(region_reference_start, dtype=np.uint64)
    assert len(hap_cigars) == nh and len(original_cigars) == n and len(alignments) == n
    cap = np.full(n, 16 if capacity is None else int(capacity), np.int64)
    i64p = C.POINTER(C.c_int64)
    for _attempt in range(2):
        out_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
        out, n_out, pos, status = np.zeros(int(out_off[-1]), np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64), np.zeros(n, np.int32)
        code = engine.lib.phmm_project_to_reference(
            engine._h, batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p), _p(batch.read_off, _lib.u32p),
            _p(batch.read_bases, _lib.u8p), _p(batch.hap_off, _lib.u32p), _p(batch.hap_bases, _lib.u8p), _p(rrh, _i32p), _p(rs, _lib.u64p),
            _p(hc_off, _lib.u32p), _p(hc, _lib.u32p), _p(hs, _lib.u32p), _p(best, _i32p), _p(sw_off, _lib.u64p), _p(sw, _lib.u32p),
            _p(sw_n, _lib.u32p), _p(sw_offset, _i32p), _p(oc_off, _lib.u32p), _p(oc, _lib.u32p), _p(out_off, _lib.u64p), _p(out, _lib.u32p),
            _p(n_out, _lib.u32p), _p(pos, i64p), _p(status, _i32p))
        if code == _lib.PHMM_ERR_CIGAR_CAPACITY:
            cap = np.maximum(cap, n_out.astype(np.int64))
            continue
        if code != _lib.PHMM_OK:
            raise PhmmError(code, engine.last_error())
        return ProjectedReads(status, pos, [out[int(out_off[r]):int(out_off[r]) + int(n_out[r])] for r in range(n)])
    raise PhmmError(_lib.PHMM_ERR_CIGAR_CAPACITY, engine.last_error())


def realign_reads(engine, batch, likelihoods, hap_cigars, hap_start_wrt_ref, region_ref_hap, region_reference_start, original_cigars,
                  hap_priority=None, keep=None, parameters=ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS,
                  overhang_strategy=OverhangStrategy.SoftClip, threshold=LOG_10_INFORMATIVE_THRESHOLD, capacity=None):
    """realign_reads_to_their_best_haplotype in one call (phmm_realign_reads): best alleles, alignments to them and their
    projection onto the reference -> (BestAlleles, ProjectedReads)."""
    n, nh = batch.n_reads, batch.n_haps
    lk = np.ascontiguousarray(likelihoods, dtype=np.float64)
    pri = None if hap_priority is None else np.ascontiguousarray(hap_priority, dtype=np.int32)
    kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    st = OverhangStrategy.NAMES[overhang_strategy] if isinstance(overhang_strategy, str) else int(overhang_strategy)
    hc_off = np.concatenate([[0], np.cumsum([len(c) for c in hap_cigars])]).astype(np.uint32)
    hc = np.concatenate([np.zeros(0, np.uint32)] + [np.asarray(c, np.uint32) for c in hap_cigars]).astype(np.uint32)
    oc_off = np.concatenate([[0], np.cumsum([len(c) for c in original_cigars])]).astype(np.uint32)
    oc = np.concatenate([np.zeros(0, np.uint32)] + [np.asarray(c, np.uint32) for c in original_cigars]).astype(np.uint32)
    hs = np.ascontiguousarray(hap_start_wrt_ref, dtype=np.uint32)
    rrh = np.ascontiguousarray(region_ref_hap, dtype=np.int32)
    rs = np.ascontiguousarray(region_reference_start, dtype=np.uint64)
    assert len(hap_cigars) == nh and len(original_cigars) == n
    best, out_lk, conf = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
    cap = np.full(n, 16 if capacity is None else int(capacity), np.int64)
    prm = parameters.as_struct()
    i64p = C.POINTER(C.c_int64)
    for _attempt in range(2):
        out_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
        out, n_out, pos, status = np.zeros(int(out_off[-1]), np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64), np.zeros(n, np.int32)
        code = engine.lib.phmm_realign_reads(
            engine._h, batch.n_regions, _p(batch.region_read_off, _lib.u32p), _p(batch.region_hap_off, _lib.u32p), _p(batch.read_off, _lib.u32p),
            _p(batch.read_bases, _lib.u8p), _p(batch.hap_off, _lib.u32p), _p(batch.hap_bases, _lib.u8p), _p(batch.out_off, _lib.u64p),
            _p(lk, _lib.f64p), _p(kp, _lib.u8p), _p(pri, _i32p), float(threshold), C.byref(prm), st, _p(rrh, _i32p), _p(rs, _lib.u64p),
            _p(hc_off, _lib.u32p), _p(hc, _lib.u32p), _p(hs, _lib.u32p), _p(oc_off, _lib.u32p), _p(oc, _lib.u32p), _p(out_off, _lib.u64p),
            _p(out, _lib.u32p), _p(n_out, _lib.u32p), _p(pos, i64p), _p(status, _i32p), _p(best, _i32p), _p(out_lk, _lib.f64p),
            _p(conf, _lib.f64p))
        if code == _lib.PHMM_ERR_CIGAR_CAPACITY:
            cap = np.maximum(cap, n_out.astype(np.int64))
            continue
        if code != _lib.PHMM_OK:
            raise PhmmError(code, engine.last_error())
        return BestAlleles(best, out_lk, conf), ProjectedReads(status, pos, [out[int(out_off[r]):int(out_off[r]) + int(n_out[r])] for r in range(n)])
    raise PhmmError(_lib.PHMM_ERR_CIGAR_CAPACITY, engine.last_error())
