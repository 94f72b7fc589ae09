"""ctypes binding of the CPU oracle (oracle/pairhmm_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from lorikeet_amd/ (the product path has no CPU fallback).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)


def build(native=False):
    """(Re)build liboracle.so with gcc.  native=True adds -march=native (bench cpu_baseline on the
    box it runs on); the default build is portable so the .so made here runs on the GPU box."""
    target = "liboracle_native.so" if native else "liboracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
    return os.path.join(_HERE, target)


def _load(path):
    lib = C.CDLL(path)
    lib.oracle_qual_to_error_prob.restype = C.c_double
    lib.oracle_qual_to_error_prob.argtypes = [C.c_uint8]
    lib.oracle_qual_to_prob.restype = C.c_double
    lib.oracle_qual_to_prob.argtypes = [C.c_uint8]
    lib.oracle_approximate_log10_sum_log10.restype = C.c_double
    lib.oracle_approximate_log10_sum_log10.argtypes = [C.c_double, C.c_double]
    lib.oracle_match_to_match_prob.restype = C.c_double
    lib.oracle_match_to_match_prob.argtypes = [C.c_uint, C.c_uint]
    lib.oracle_qual_to_trans_probs.restype = None
    lib.oracle_qual_to_trans_probs.argtypes = [_f64p, C.c_uint8, C.c_uint8, C.c_uint8]
    lib.oracle_pairhmm_new.restype = C.c_void_p
    lib.oracle_pairhmm_new.argtypes = [C.c_size_t, C.c_size_t]
    lib.oracle_pairhmm_free.restype = None
    lib.oracle_pairhmm_free.argtypes = [C.c_void_p]
    lib.oracle_pairhmm_do_not_use_tristate_correction.restype = None
    lib.oracle_pairhmm_do_not_use_tristate_correction.argtypes = [C.c_void_p]
    lib.oracle_find_first_position_where_haplotypes_differ.restype = C.c_size_t
    lib.oracle_find_first_position_where_haplotypes_differ.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t]
    lib.oracle_compute_read_likelihood_given_haplotype_log10.restype = C.c_double
    lib.oracle_compute_read_likelihood_given_haplotype_log10.argtypes = [
        C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, _u8p, _u8p, _u8p, C.c_int, _u8p, C.c_size_t,
        C.POINTER(C.c_int)]
    lib.oracle_compute.restype = C.c_int
    lib.oracle_compute.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, _u8p, _u8p, _u8p, _u8p, _u8p, _u32p, _u8p,
                                   _u64p, _f64p, C.c_int, C.c_int]
    lib.oracle_simd_compute.restype = C.c_int
    lib.oracle_simd_compute.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, _u8p, _u8p, _u8p, _u8p, _u8p, _u32p, _u8p,
                                        _u64p, _f64p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    lib.oracle_simd_lanes.restype = C.c_int
    lib.oracle_sw_align.restype = C.c_int
    lib.oracle_sw_align.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int,
                                    _u32p, C.POINTER(C.c_int32)]
    lib.oracle_mm_table_len.restype = C.c_size_t
    lib.oracle_mm_prob_table.restype = _f64p
    # engine_oracle.c
    lib.oracle_pcr_error_model_cache.restype = None
    lib.oracle_pcr_error_model_cache.argtypes = [C.c_int, _u8p]
    lib.oracle_find_tandem_repeat_length.restype = C.c_size_t
    lib.oracle_find_tandem_repeat_length.argtypes = [_u8p, C.c_size_t, C.c_size_t]
    lib.oracle_find_number_of_repetitions.restype = C.c_size_t
    lib.oracle_find_number_of_repetitions.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int]
    lib.oracle_find_number_of_repetitions_main.restype = C.c_size_t
    lib.oracle_find_number_of_repetitions_main.argtypes = [_u8p, C.c_size_t, C.c_size_t, _u8p, C.c_size_t, C.c_size_t, C.c_int]
    lib.oracle_modify_read_qualities.restype = None
    lib.oracle_modify_read_qualities.argtypes = [C.c_int, _u8p, C.c_size_t, C.c_uint8, _u8p, _u8p, _u8p, C.c_uint8, C.c_int]
    lib.oracle_read_disqualification_threshold.restype = C.c_double
    lib.oracle_read_disqualification_threshold.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_double, C.c_double]
    lib.oracle_log10_min_true_likelihood.restype = C.c_double
    lib.oracle_log10_min_true_likelihood.argtypes = [C.c_size_t, C.c_double, C.c_int]
    lib.oracle_log10_dynamic_read_qual_threshold.restype = C.c_double
    lib.oracle_log10_dynamic_read_qual_threshold.argtypes = [_u8p, C.c_size_t, C.c_double]
    lib.oracle_normalize_likelihoods.restype = None
    lib.oracle_normalize_likelihoods.argtypes = [_f64p, C.c_size_t, C.c_size_t, C.c_double, C.c_int, C.c_long]
    lib.oracle_filter_poorly_modeled_evidence.restype = C.c_size_t
    lib.oracle_filter_poorly_modeled_evidence.argtypes = [_f64p, C.c_size_t, C.c_size_t, _f64p, _u8p]
    _i32p, _i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    lib.oracle_cigar_builder.restype = C.c_int
    lib.oracle_cigar_builder.argtypes = [_u32p, C.c_uint32, C.c_int, C.c_int, _u32p, C.c_uint32, _u32p, _u32p, _u32p]
    lib.oracle_consolidated_padded_cigar.restype = C.c_int
    lib.oracle_consolidated_padded_cigar.argtypes = [_u32p, C.c_uint32, C.c_uint32, _u32p, C.c_uint32, _u32p]
    lib.oracle_read_start_on_reference_haplotype.restype = C.c_int
    lib.oracle_read_start_on_reference_haplotype.argtypes = [_u32p, C.c_uint32, C.c_uint32, _u32p]
    lib.oracle_trim_cigar.restype = C.c_int
    lib.oracle_trim_cigar.argtypes = [_u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _u32p, C.c_uint32, _u32p, _u32p, _u32p]
    lib.oracle_apply_cigar_to_cigar.restype = C.c_int
    lib.oracle_apply_cigar_to_cigar.argtypes = [_u32p, C.c_uint32, _u32p, C.c_uint32, _u32p, C.c_uint32, _u32p]
    lib.oracle_left_align_indels.restype = C.c_int
    lib.oracle_left_align_indels.argtypes = [_u32p, C.c_uint32, _u8p, C.c_uint32, _u8p, C.c_uint32, C.c_uint32, _u32p, C.c_uint32, _u32p,
                                             _u32p, _u32p]
    lib.oracle_append_clipped_elements.restype = C.c_int
    lib.oracle_append_clipped_elements.argtypes = [_u32p, C.c_uint32, _u32p, C.c_uint32, _u32p, C.c_uint32, _u32p]
    lib.oracle_create_read_aligned_to_ref.restype = C.c_int
    lib.oracle_create_read_aligned_to_ref.argtypes = [_u32p, C.c_uint32, C.c_int32, _u32p, C.c_uint32, C.c_uint32, C.c_uint64, _u8p, C.c_uint32,
                                                      _u8p, C.c_uint32, _u32p, C.c_uint32, C.c_uint32, _i64p, _u32p, C.c_uint32, _u32p]
    lib.oracle_calculate_cigar.restype = C.c_int
    lib.oracle_calculate_cigar.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int, _u32p,
                                           C.c_uint32, _u32p]
    lib.oracle_best_alleles.restype = None
    lib.oracle_best_alleles.argtypes = [_f64p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int32), C.c_double, C.POINTER(C.c_int32), _f64p, _f64p]
    return lib


_lib = None


def lib(native=False):
    global _lib
    if native:
        return _load(build(native=True))
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = _load(path)
    return _lib


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                             dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


class OraclePairHMM:
    """Mirror of the reference's scalar PairHMM object (pair_hmm.rs:128-165, 405-501)."""

    def __init__(self, max_read_length, max_haplotype_length):
        self._lib = lib()
        self._h = self._lib.oracle_pairhmm_new(max_read_length, max_haplotype_length)

    def do_not_use_tristate_correction(self):
        self._lib.oracle_pairhmm_do_not_use_tristate_correction(self._h)

    def compute_read_likelihood_given_haplotype_log10(self, hap, read, quals, ins, dele, gcp, recache=True,
                                                      next_hap=None):
        hap_a, hap_p = _u8(hap)
        read_a, read_p = _u8(read)
        q_a, q_p = _u8(quals)
        i_a, i_p = _u8(ins)
        d_a, d_p = _u8(dele)
        g_a, g_p = _u8(gcp)
        assert len(q_a) == len(read_a) == len(i_a) == len(d_a) == len(g_a)
        if next_hap is None:
            n_a, n_p, n_len = None, None, 0
        else:
            n_a, n_p = _u8(next_hap)
            n_len = len(n_a)
        st = C.c_int(0)
        v = self._lib.oracle_compute_read_likelihood_given_haplotype_log10(
            self._h, hap_p, len(hap_a), read_p, len(read_a), q_p, i_p, d_p, g_p, int(bool(recache)), n_p, n_len,
            C.byref(st))
        if st.value not in (0,):
            raise AssertionError({1: "Must call initialize first", 2: "Haplotype bases is too long",
                                  4: "PairHmm Log Probability cannot be greater than 0.0"}.get(st.value, "error"))
        return v

    def __del__(self):
        try:
            self._lib.oracle_pairhmm_free(self._h)
        except Exception:
            pass


def compute_batch(batch, disable_tristate=False, n_threads=1, native=False):
    """batch: dict with the SoA arrays of include/phmm.h (numpy).  Returns out (float64)."""
    L = lib(native=native)
    rro = np.ascontiguousarray(batch["region_read_off"], dtype=np.uint32)
    rho = np.ascontiguousarray(batch["region_hap_off"], dtype=np.uint32)
    ro = np.ascontiguousarray(batch["read_off"], dtype=np.uint32)
    ho = np.ascontiguousarray(batch["hap_off"], dtype=np.uint32)
    oo = np.ascontiguousarray(batch["out_off"], dtype=np.uint64)
    arrs = [np.ascontiguousarray(batch[k], dtype=np.uint8) for k in
            ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")]
    out = np.zeros(int(oo[-1]), dtype=np.float64)
    p8 = [a.ctypes.data_as(_u8p) for a in arrs]
    st = L.oracle_compute(len(rro) - 1, rro.ctypes.data_as(_u32p), rho.ctypes.data_as(_u32p),
                          ro.ctypes.data_as(_u32p), p8[0], p8[1], p8[2], p8[3], p8[4], ho.ctypes.data_as(_u32p),
                          p8[5], oo.ctypes.data_as(_u64p), out.ctypes.data_as(_f64p), int(disable_tristate),
                          int(n_threads))
    if st:
        raise AssertionError("oracle status %d" % st)
    return out


def compute_batch_simd(batch, disable_tristate=False, n_threads=1, native=False):
    """The stand-in for the reference's VECTOR arm (oracle/pairhmm_simd.c: f32 first, f64 redo, one SIMD lane per
    haplotype).  Returns (out float64, number of pairs redone in f64)."""
    L = lib(native=native)
    rro = np.ascontiguousarray(batch["region_read_off"], dtype=np.uint32)
    rho = np.ascontiguousarray(batch["region_hap_off"], dtype=np.uint32)
    ro = np.ascontiguousarray(batch["read_off"], dtype=np.uint32)
    ho = np.ascontiguousarray(batch["hap_off"], dtype=np.uint32)
    oo = np.ascontiguousarray(batch["out_off"], dtype=np.uint64)
    arrs = [np.ascontiguousarray(batch[k], dtype=np.uint8) for k in
            ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")]
    out = np.zeros(int(oo[-1]), dtype=np.float64)
    p8 = [a.ctypes.data_as(_u8p) for a in arrs]
    redone = C.c_uint64(0)
    L.oracle_simd_compute(len(rro) - 1, rro.ctypes.data_as(_u32p), rho.ctypes.data_as(_u32p), ro.ctypes.data_as(_u32p),
                          p8[0], p8[1], p8[2], p8[3], p8[4], ho.ctypes.data_as(_u32p), p8[5], oo.ctypes.data_as(_u64p),
                          out.ctypes.data_as(_f64p), int(disable_tristate), int(n_threads), C.byref(redone))
    return out, int(redone.value)


def load_kat(path):
    """Parse the reference fixture pairhmm-testdata.txt the way
    tests/vector_pair_hmm_unit_tests.rs:22-50 does (ASCII-33; base quals floored at 6)."""
    rows = []
    with open(path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            t = line.split()
            hap, read = t[0].encode(), t[1].encode()

            def q(s, lo):
                return np.maximum(np.frombuffer(s.encode(), dtype=np.uint8).astype(np.int32) - 33, lo).astype(np.uint8)
            rows.append(dict(hap=hap, read=read, qual=q(t[2], 6), ins=q(t[3], 0), dele=q(t[4], 0), gcp=q(t[5], 0),
                             expected=float(t[6])))
    return rows


# ---- engine-level steps (engine_oracle.c) -----------------------------------------------------------
PCR_MODELS = {"none": 0, "hostile": 1, "aggressive": 2, "conservative": 3}


def pcr_error_model_cache(model):
    c = np.zeros(101, np.uint8)
    lib().oracle_pcr_error_model_cache(PCR_MODELS[model], c.ctypes.data_as(_u8p))
    return c


def find_number_of_repetitions(repeat_unit, test_string, leading_repeats):
    """VariantContextUtils::find_number_of_repetitions (variant_context_utils.rs:240-257)."""
    (u, up), (t, tp) = _u8(repeat_unit), _u8(test_string)
    return int(lib().oracle_find_number_of_repetitions(up, len(u), tp, len(t), int(leading_repeats)))


def find_number_of_repetitions_main(repeat_unit_full, offset_in_repeat_unit_full, repeat_unit_length, test_string_full,
                                    offset_in_test_string_full, test_string_length, leading_repeats):
    """VariantContextUtils::find_number_of_repetitions_main (variant_context_utils.rs:276-335)."""
    (u, up), (t, tp) = _u8(repeat_unit_full), _u8(test_string_full)
    return int(lib().oracle_find_number_of_repetitions_main(up, offset_in_repeat_unit_full, repeat_unit_length, tp,
                                                            offset_in_test_string_full, test_string_length, int(leading_repeats)))


def modify_read_qualities(model, bases, mapq, quals, ins, dele, base_quality_score_threshold,
                          disable_cap_read_qualities_to_mapq=False):
    """Returns modified copies (quals, ins, dele) -- engine.rs:352-388."""
    b, bp = _u8(bases)
    q = np.array(quals, np.uint8); i = np.array(ins, np.uint8); d = np.array(dele, np.uint8)
    lib().oracle_modify_read_qualities(PCR_MODELS[model], bp, len(b), int(mapq), q.ctypes.data_as(_u8p),
                                       i.ctypes.data_as(_u8p), d.ctypes.data_as(_u8p),
                                       int(base_quality_score_threshold), int(disable_cap_read_qualities_to_mapq))
    return q, i, d


def read_disqualification_threshold(orig_quals, dynamic, scale, expected_error_rate_per_base):
    q, qp = _u8(np.asarray(orig_quals, np.uint8))
    return lib().oracle_read_disqualification_threshold(qp, len(q), int(dynamic), float(scale),
                                                       float(expected_error_rate_per_base))


def normalize_likelihoods(values, cap, symmetric, reference_allele_index):
    """values: [allele, read] float64 (modified in place and returned)."""
    v = np.ascontiguousarray(values, np.float64)
    lib().oracle_normalize_likelihoods(v.ctypes.data_as(_f64p), v.shape[0], v.shape[1], float(cap), int(symmetric),
                                       -1 if reference_allele_index is None else int(reference_allele_index))
    return v


def filter_poorly_modeled_evidence(values, thresholds):
    v = np.ascontiguousarray(values, np.float64)
    t = np.ascontiguousarray(thresholds, np.float64)
    keep = np.zeros(v.shape[1], np.uint8)
    n = lib().oracle_filter_poorly_modeled_evidence(v.ctypes.data_as(_f64p), v.shape[0], v.shape[1],
                                                    t.ctypes.data_as(_f64p), keep.ctypes.data_as(_u8p))
    return v, keep.astype(bool), int(n)


# ---- Smith-Waterman (oracle/sw_oracle.c) -----------------------------------------------------------------------------
SW_STRATEGIES = {"SoftClip": 0, "InDel": 1, "LeadingInDel": 2, "Ignore": 3}   # the numbering of include/phmm.h
_CIGAR_OPS = "MIDNSHP=X"


def cigar_to_string(elements):
    """[(len << 4) | BAM op, ...] -> '5M3S'."""
    return "".join("%d%s" % (int(e) >> 4, _CIGAR_OPS[int(e) & 15]) for e in elements)


def sw_align(reference, alternate, params, strategy):
    """The reference's scalar SmithWatermanAligner::align (smith_waterman_aligner.rs:47-107).  params = (match,
    mismatch, gap open, gap extend); strategy = name or number.  Returns (cigar elements uint32[], alignment offset)."""
    L = lib()
    r = np.frombuffer(bytes(reference), np.uint8) if isinstance(reference, (bytes, bytearray, str)) and not isinstance(reference, str) \
        else np.ascontiguousarray(np.frombuffer(reference.encode(), np.uint8) if isinstance(reference, str) else reference, dtype=np.uint8)
    a = np.frombuffer(bytes(alternate), np.uint8) if isinstance(alternate, (bytes, bytearray)) \
        else np.ascontiguousarray(np.frombuffer(alternate.encode(), np.uint8) if isinstance(alternate, str) else alternate, dtype=np.uint8)
    st = SW_STRATEGIES[strategy] if isinstance(strategy, str) else int(strategy)
    cig = np.zeros(len(r) + len(a) + 3, np.uint32)
    off = C.c_int32(0)
    n = L.oracle_sw_align(r.ctypes.data_as(_u8p), len(r), a.ctypes.data_as(_u8p), len(a), int(params[0]), int(params[1]),
                          int(params[2]), int(params[3]), st, cig.ctypes.data_as(_u32p), C.byref(off))
    if n < 0:
        raise AssertionError("non-empty sequences are required for the Smith-Waterman calculation")
    return cig[:n].copy(), int(off.value)


def best_alleles(values, priorities=None, threshold=0.2):
    """AlleleLikelihoods::best_alleles_tie_breaking for one sample (src/model/allele_likelihoods.rs:457-554, :1069-1095):
    `values` [allele, read] -> (best allele index per read, its likelihood, confidence)."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    na, nr = v.shape
    pri = None if priorities is None else np.ascontiguousarray(priorities, dtype=np.int32)
    best, lk, conf = np.zeros(nr, np.int32), np.zeros(nr), np.zeros(nr)
    lib().oracle_best_alleles(v.ctypes.data_as(_f64p), na, nr, None if pri is None else pri.ctypes.data_as(C.POINTER(C.c_int32)),
                              threshold, best.ctypes.data_as(C.POINTER(C.c_int32)), lk.ctypes.data_as(_f64p), conf.ctypes.data_as(_f64p))
    return best, lk, conf


# ---- CIGAR algebra (oracle/cigar_oracle.c): src/reads/cigar_builder.rs, src/reads/alignment_utils.rs:60-566 ----------------------
_CIGAR_OPS = "MIDNSHP=X"


def _seq(a):
    if isinstance(a, str):
        a = a.encode()
    return np.ascontiguousarray(np.frombuffer(bytes(a), dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8)


def parse_cigar(text):
    """'3M2D4M' -> BAM-encoded elements, (length << 4) | op."""
    import re
    return np.array([(int(n) << 4) | _CIGAR_OPS.index(o) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", text)], np.uint32)


def _elems(c):
    return parse_cigar(c) if isinstance(c, str) else np.ascontiguousarray(c, dtype=np.uint32)


def _pu32(a):
    return a.ctypes.data_as(_u32p)


class CigarError(Exception):
    """Where the reference returns Err or panics; .code is the status of oracle/cigar_oracle.c."""

    def __init__(self, code):
        super().__init__("cigar oracle status %d" % code)
        self.code = code


def cigar_builder(elements, remove_deletions_at_ends=True, allow_empty=False):
    """CigarBuilder::new(remove).add(e)... .make(allow_empty) -> (cigar string, leading, trailing deletion bases removed)."""
    e = np.concatenate([_elems(x) for x in elements]) if len(elements) else np.zeros(0, np.uint32)
    out, n, lead, trail = np.zeros(len(e) + 4, np.uint32), C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = lib().oracle_cigar_builder(_pu32(e), len(e), int(remove_deletions_at_ends), int(allow_empty), _pu32(out), len(out), C.byref(n),
                                    C.byref(lead), C.byref(trail))
    if st:
        raise CigarError(st)
    return cigar_to_string(out[:n.value]), lead.value, trail.value


def consolidated_padded_cigar(cigar, pad_size):
    """Haplotype::get_consolidated_padded_cigar (haplotype.rs:248-256) -> cigar string."""
    c = _elems(cigar)
    out, n = np.zeros(len(c) + 4, np.uint32), C.c_uint32()
    st = lib().oracle_consolidated_padded_cigar(_pu32(c), len(c), int(pad_size), _pu32(out), len(out), C.byref(n))
    if st:
        raise CigarError(st)
    return cigar_to_string(out[:n.value])


def read_start_on_reference_haplotype(cigar, read_start_on_haplotype):
    c, out = _elems(cigar), C.c_uint32()
    st = lib().oracle_read_start_on_reference_haplotype(_pu32(c), len(c), read_start_on_haplotype, C.byref(out))
    if st:
        raise CigarError(st)
    return out.value


def trim_cigar(cigar, start, end, by_reference):
    """AlignmentUtils::trim_cigar_by_reference / trim_cigar_by_bases -> (cigar string, leading, trailing removed)."""
    c = _elems(cigar)
    out, n, lead, trail = np.zeros(len(c) + 4, np.uint32), C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = lib().oracle_trim_cigar(_pu32(c), len(c), start, end, int(by_reference), _pu32(out), len(out), C.byref(n), C.byref(lead), C.byref(trail))
    if st:
        raise CigarError(st)
    return cigar_to_string(out[:n.value]), lead.value, trail.value


def apply_cigar_to_cigar(first_to_second, second_to_third):
    a, b = _elems(first_to_second), _elems(second_to_third)
    out, n = np.zeros(4096, np.uint32), C.c_uint32()
    st = lib().oracle_apply_cigar_to_cigar(_pu32(a), len(a), _pu32(b), len(b), _pu32(out), len(out), C.byref(n))
    if st:
        raise CigarError(st)
    return cigar_to_string(out[:n.value])


def left_align_indels(cigar, ref, read, read_start):
    c, r, q = _elems(cigar), _seq(ref), _seq(read)
    out, n, lead, trail = np.zeros(4 * len(c) + 8, np.uint32), C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = lib().oracle_left_align_indels(_pu32(c), len(c), r.ctypes.data_as(_u8p), len(r), q.ctypes.data_as(_u8p), len(q), read_start,
                                        _pu32(out), len(out), C.byref(n), C.byref(lead), C.byref(trail))
    if st:
        raise CigarError(st)
    return cigar_to_string(out[:n.value]), lead.value, trail.value


def append_clipped_elements(cigar, original):
    a, b = _elems(cigar), _elems(original)
    out, n = np.zeros(len(a) + len(b) + 4, np.uint32), C.c_uint32()
    st = lib().oracle_append_clipped_elements(_pu32(a), len(a), _pu32(b), len(b), _pu32(out), len(out), C.byref(n))
    if st:
        raise CigarError(st)
    return cigar_to_string(out[:n.value])


def create_read_aligned_to_ref(sw_cigar, sw_offset, hap_cigar, hap_start_wrt_ref, reference_start, ref_bases, read, original_cigar,
                               original_read_len=None):
    """create_read_aligned_to_ref from the alignment on (src/reads/alignment_utils.rs:60-165) -> None (read unchanged) or
    (new position, new cigar string)."""
    a, hc, oc, r, q = _elems(sw_cigar), _elems(hap_cigar), _elems(original_cigar), _seq(ref_bases), _seq(read)
    out, n, pos = np.zeros(4 * (len(a) + len(hc)) + len(oc) + 16, np.uint32), C.c_uint32(), C.c_int64()
    st = lib().oracle_create_read_aligned_to_ref(_pu32(a), len(a), int(sw_offset), _pu32(hc), len(hc), int(hap_start_wrt_ref), int(reference_start),
                                                 r.ctypes.data_as(_u8p), len(r), q.ctypes.data_as(_u8p), len(q), _pu32(oc), len(oc),
                                                 len(q) if original_read_len is None else int(original_read_len), C.byref(pos), _pu32(out),
                                                 len(out), C.byref(n))
    if st == 1:
        return None
    if st:
        raise CigarError(st)
    return pos.value, cigar_to_string(out[:n.value])


def calculate_cigar(ref_seq, alt_seq, parameters=(200, -150, -260, -11), strategy="InDel"):
    """CigarUtils::calculate_cigar (src/reads/cigar_utils.rs:358-457) -> cigar string, or None (is_s_w_failure)."""
    r, a = _seq(ref_seq), _seq(alt_seq)
    out, n = np.zeros(len(r) + len(a) + 48, np.uint32), C.c_uint32()
    st = lib().oracle_calculate_cigar(r.ctypes.data_as(_u8p), len(r), a.ctypes.data_as(_u8p), len(a), *[int(x) for x in parameters],
                                      SW_STRATEGIES[strategy] if isinstance(strategy, str) else int(strategy), _pu32(out), len(out), C.byref(n))
    if st == 1:
        return None
    if st:
        raise CigarError(st)
    return cigar_to_string(out[:n.value])
