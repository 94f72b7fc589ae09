"""Host-side mirror of the reference's Smith-Waterman surface (src/smith_waterman/smith_waterman_aligner.rs) over
`phmm_sw_align`: same names, argument meaning and error behaviour; every alignment runs on the MI355X (no CPU path).

    SmithWatermanAligner.align(reference, alternate, parameters, overhang_strategy) -> SmithWatermanAlignmentResult
    SmithWatermanAligner.align_batch(pairs, parameters, overhang_strategy)           -> [SmithWatermanAlignmentResult]
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import HipPairHMMEngine, PhmmError


class OverhangStrategy:
    """gkl::smithwaterman::OverhangStrategy (numbering of include/phmm.h)."""
    SoftClip, InDel, LeadingInDel, Ignore = 0, 1, 2, 3
    NAMES = {"SoftClip": 0, "InDel": 1, "LeadingInDel": 2, "Ignore": 3}


class Parameters:
    """gkl::smithwaterman::Parameters::new(match_value, mismatch_penalty, gap_open_penalty, gap_extend_penalty)."""

    def __init__(self, match_value, mismatch_penalty, gap_open_penalty, gap_extend_penalty):
        self.match_value, self.mismatch_penalty = int(match_value), int(mismatch_penalty)
        self.gap_open_penalty, self.gap_extend_penalty = int(gap_open_penalty), int(gap_extend_penalty)

    def as_struct(self):
        return _lib.SwParameters(self.match_value, self.mismatch_penalty, self.gap_open_penalty, self.gap_extend_penalty)


# smith_waterman_aligner.rs:11-26
ORIGINAL_DEFAULT = Parameters(3, -1, -4, -3)
STANDARD_NGS = Parameters(25, -50, -110, -6)
NEW_SW_PARAMETERS = Parameters(200, -150, -260, -11)
ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS = Parameters(10, -15, -30, -5)

_OPS = "MIDNSHP=X"


class SmithWatermanAlignmentResult:
    """cigar: list of (length, op char); alignment_offset: int (smith_waterman_aligner.rs:454-476)."""

    def __init__(self, elements, alignment_offset):
        self.elements = np.asarray(elements, np.uint32)
        self.alignment_offset = int(alignment_offset)

    def get_alignment_offset(self):
        return self.alignment_offset

    def get_cigar(self):
        return [(int(e) >> 4, _OPS[int(e) & 15]) for e in self.elements]

    def cigar_string(self):
        return "".join("%d%s" % c for c in self.get_cigar())

    def __eq__(self, o):
        return self.alignment_offset == o.alignment_offset and np.array_equal(self.elements, o.elements)

    def __repr__(self):
        return "SmithWatermanAlignmentResult(%s @ %d)" % (self.cigar_string(), self.alignment_offset)


def _u8(x):
    if isinstance(x, str):
        x = x.encode()
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


class SmithWatermanAligner:
    """One aligner == one engine handle (created on first use when none is given)."""

    def __init__(self, engine=None, device_id=0):
        self.engine = engine or HipPairHMMEngine(device_id)

    def align(self, reference, alternate, parameters, overhang_strategy):
        return self.align_batch([(reference, alternate)], parameters, overhang_strategy)[0]

    def align_batch(self, pairs, parameters, overhang_strategy, capacity=None):
        """pairs: iterable of (reference, alternate).  capacity: CIGAR elements reserved per alignment (default 24;
        alignments that need more are redone with what they need -- the library reports the size)."""
        pairs = list(pairs)
        return self.align_indexed([r for r, _ in pairs], [a for _, a in pairs], None, parameters, overhang_strategy, capacity)

    def align_indexed(self, references, alternates, ref_index, parameters, overhang_strategy, capacity=None):
        """alternates[a] against references[ref_index[a]] (phmm_sw_align_indexed: shared references cross the bus once);
        ref_index None = one reference per alternate; an index of -1 skips the alignment (None in the result)."""
        refs = [_u8(r) for r in references]
        alts = [_u8(a) for a in alternates]
        # smith_waterman_aligner.rs:65-68 asserts on what it aligns: a reference no alignment names, or the read of a
        # skipped alignment, may be empty
        used = set(range(len(refs))) if ref_index is None else {int(i) for i in ref_index if int(i) >= 0}
        for r in [refs[i] for i in sorted(used) if i < len(refs)] + [a for k, a in enumerate(alts) if ref_index is None or int(ref_index[k]) >= 0]:
            if len(r) == 0:
                raise AssertionError("non-empty sequences are required for the Smith-Waterman calculation")
        st = OverhangStrategy.NAMES[overhang_strategy] if isinstance(overhang_strategy, str) else int(overhang_strategy)
        n = len(alts)
        if n == 0:
            return []
        idx = None
        if ref_index is not None:
            idx = np.where(np.asarray(ref_index, np.int64) < 0, _lib.PHMM_SW_NO_REFERENCE, np.asarray(ref_index, np.int64)).astype(np.uint32)
        ref_off = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.uint32)
        alt_off = np.concatenate([[0], np.cumsum([len(a) for a in alts])]).astype(np.uint32)
        rb, ab = np.ascontiguousarray(np.concatenate(refs + [np.zeros(1, np.uint8)])), np.ascontiguousarray(np.concatenate(alts + [np.zeros(1, np.uint8)]))
        cap = np.full(n, 24 if capacity is None else int(capacity), np.int64)
        prm = parameters.as_struct()
        eng = self.engine
        for _attempt in range(2):
            cig_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
            cigar = np.zeros(int(cig_off[-1]), np.uint32)
            n_cig = np.zeros(n, np.uint32)
            off = np.zeros(n, np.int32)
            tail = (alt_off.ctypes.data_as(_lib.u32p), ab.ctypes.data_as(_lib.u8p), C.byref(prm), st,
                    cig_off.ctypes.data_as(_lib.u64p), cigar.ctypes.data_as(_lib.u32p), n_cig.ctypes.data_as(_lib.u32p),
                    off.ctypes.data_as(C.POINTER(C.c_int32)))
            if idx is None:
                code = eng.lib.phmm_sw_align(eng._h, n, ref_off.ctypes.data_as(_lib.u32p), rb.ctypes.data_as(_lib.u8p), *tail)
            else:
                code = eng.lib.phmm_sw_align_indexed(eng._h, len(refs), ref_off.ctypes.data_as(_lib.u32p), rb.ctypes.data_as(_lib.u8p), n,
                                                     idx.ctypes.data_as(_lib.u32p), *tail)
            if code == _lib.PHMM_ERR_CIGAR_CAPACITY:
                cap = np.maximum(cap, n_cig.astype(np.int64))
                continue
            if code != _lib.PHMM_OK:
                raise PhmmError(code, eng.last_error())
            return [None if idx is not None and idx[a] == _lib.PHMM_SW_NO_REFERENCE else
                    SmithWatermanAlignmentResult(cigar[int(cig_off[a]):int(cig_off[a]) + int(n_cig[a])], off[a]) for a in range(n)]
        raise PhmmError(_lib.PHMM_ERR_CIGAR_CAPACITY, eng.last_error())


def calculate_cigar(engine, pairs, parameters=NEW_SW_PARAMETERS, overhang_strategy=OverhangStrategy.InDel, capacity=None):
    """CigarUtils::calculate_cigar (src/reads/cigar_utils.rs:358-457) for a batch of (reference, haplotype) pairs
    (phmm_calculate_cigar) -> per pair the BAM-encoded elements, or None where the reference returns None (is_s_w_failure).
    Raises PhmmError where the reference would panic."""
    refs = [_u8(r) for r, _ in pairs]
    alts = [_u8(a) for _, a in pairs]
    n = len(refs)
    if n == 0:
        return []
    st = OverhangStrategy.NAMES[overhang_strategy] if isinstance(overhang_strategy, str) else int(overhang_strategy)
    ref_off = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.uint32)
    alt_off = np.concatenate([[0], np.cumsum([len(a) for a in alts])]).astype(np.uint32)
    rb = np.ascontiguousarray(np.concatenate(refs + [np.zeros(0, np.uint8)]))
    ab = np.ascontiguousarray(np.concatenate(alts + [np.zeros(0, np.uint8)]))
    cap = np.full(n, 16 if capacity is None else int(capacity), np.int64)
    prm = parameters.as_struct()
    for _attempt in range(2):
        cig_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
        cigar, n_cig, status = np.zeros(int(cig_off[-1]), np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
        code = engine.lib.phmm_calculate_cigar(engine._h, n, ref_off.ctypes.data_as(_lib.u32p), rb.ctypes.data_as(_lib.u8p),
                                               alt_off.ctypes.data_as(_lib.u32p), ab.ctypes.data_as(_lib.u8p), C.byref(prm), st,
                                               cig_off.ctypes.data_as(_lib.u64p), cigar.ctypes.data_as(_lib.u32p),
                                               n_cig.ctypes.data_as(_lib.u32p), status.ctypes.data_as(C.POINTER(C.c_int32)))
        if code == _lib.PHMM_ERR_CIGAR_CAPACITY:
            cap = np.maximum(cap, n_cig.astype(np.int64))
            continue
        if code != _lib.PHMM_OK:
            raise PhmmError(code, engine.last_error())
        if np.any(status < 0):
            raise PhmmError(_lib.PHMM_ERR_INTERNAL, "calculate_cigar: the reference panics on pair %d (status %d)" % (int(np.argmax(status < 0)), int(status.min())))
        return [None if status[a] == 1 else cigar[int(cig_off[a]):int(cig_off[a]) + int(n_cig[a])] for a in range(n)]
    raise PhmmError(_lib.PHMM_ERR_CIGAR_CAPACITY, engine.last_error())
