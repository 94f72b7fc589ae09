"""ctypes binding of libphmm.so (the C ABI declared in include/phmm.h).

The library is built in-tree by `__graft_entry__.build()` (or `make -C lorikeet_amd/csrc`).
There is deliberately NO fallback: if the shared library is missing, or no HIP device is
present when an engine is created, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphmm.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)

PHMM_FLAG_NO_TRISTATE = 1
PHMM_FLAG_F32_FIRST = 2  # opt-in: f32 sweep first, f64 redo of what f32 cannot be trusted with (include/phmm.h)
PHMM_OK = 0
PHMM_ERR_INVALID_ARG = 1
PHMM_ERR_NO_DEVICE = 2
PHMM_ERR_HIP = 3
PHMM_ERR_POSITIVE_RESULT = 4
PHMM_ERR_NOT_BOUND = 5
PHMM_ERR_NO_MEMORY = 6
PHMM_ERR_INTERNAL = 7
PHMM_ERR_CIGAR_CAPACITY = 8
PHMM_SW_SOFTCLIP, PHMM_SW_INDEL, PHMM_SW_LEADING_INDEL, PHMM_SW_IGNORE = 0, 1, 2, 3
PHMM_SW_NO_REFERENCE = 0xffffffff
PHMM_PROJECT_REALIGNED, PHMM_PROJECT_UNCHANGED = 0, 1
PHMM_REGION_SKIP_SINGLE_ALLELE = 1

class EngineConfig(C.Structure):
    """phmm_engine_config (include/phmm.h)."""
    _fields_ = [("constant_gcp", C.c_uint8), ("pcr_error_model", C.c_uint8),
                ("base_quality_score_threshold", C.c_uint8), ("dynamic_read_disqualification", C.c_uint8),
                ("symmetrically_normalize_alleles_to_reference", C.c_uint8),
                ("disable_cap_read_qualities_to_mapq", C.c_uint8), ("reserved", C.c_uint8 * 2),
                ("log10_global_read_mismapping_rate", C.c_double), ("read_disqualification_scale", C.c_double),
                ("expected_error_rate_per_base", C.c_double)]


class SwParameters(C.Structure):
    """phmm_sw_parameters == gkl::smithwaterman::Parameters::new(match, mismatch, gap open, gap extend)."""
    _fields_ = [("match_value", C.c_int32), ("mismatch_penalty", C.c_int32), ("gap_open_penalty", C.c_int32),
                ("gap_extend_penalty", C.c_int32)]


class RealignConfig(C.Structure):
    """phmm_realign_config (include/phmm.h)."""
    _fields_ = [("sw_parameters", SwParameters), ("overhang_strategy", C.c_int32), ("flags", C.c_uint32),
                ("informative_threshold", C.c_double)]


class PlanInfo(C.Structure):
    """phmm_plan_info (include/phmm.h)."""
    _fields_ = [("cells", C.c_uint64), ("chain_cells", C.c_uint64), ("chain_items", C.c_uint64), ("n_launches", C.c_uint32),
                ("n_chain_launches", C.c_uint32), ("min_reads_per_run", C.c_uint32), ("reserved", C.c_uint32),
                ("dominant_kernel", C.c_char * 64), ("swept_cells", C.c_uint64), ("pad_column_cells", C.c_uint64),
                ("pad_slot_cells", C.c_uint64)]


_REGION_ARGS = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u8p, u8p, u8p, u8p, u32p, u32p, u8p,
                C.POINTER(C.c_int32), u64p, C.POINTER(C.c_int32), u64p, u32p, u32p, u32p, u32p, u32p, u64p, f64p, u8p,
                C.POINTER(C.c_int32), f64p, f64p, u32p, u32p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]

# every symbol include/phmm.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("phmm_device_count", C.c_int, []),
    ("phmm_create", C.c_void_p, [C.c_int, C.c_uint]),
    ("phmm_destroy", None, [C.c_void_p]),
    ("phmm_last_error", C.c_char_p, [C.c_void_p]),
    ("phmm_compute", C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u8p, u8p, u8p, u8p, u32p, u8p, u64p, f64p]),
    ("phmm_submit", C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u8p, u8p, u8p, u8p, u32p, u8p, u64p, f64p,
                              C.POINTER(C.c_uint64)]),
    ("phmm_wait", C.c_int, [C.c_void_p, C.c_uint64]),
    ("phmm_submit_stats", None, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("phmm_assign_regions", C.c_int, [C.c_uint32, u32p, u32p, u32p, u32p, C.c_uint32, u32p]),
    ("phmm_split_regions", C.c_int, [C.c_uint32, u32p, u32p, u32p, u32p, C.c_uint32, u32p]),
    ("phmm_compute_multi", C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, u32p, u32p, u32p, u8p, u8p, u8p, u8p, u8p,
                                     u32p, u8p, u64p, f64p]),
    ("phmm_batch_create", C.c_void_p, [C.c_void_p, C.c_uint32, u32p, u32p, u32p, u32p, u64p]),
    ("phmm_batch_destroy", None, [C.c_void_p]),
    ("phmm_batch_bind_device", C.c_int, [C.c_void_p] + [C.c_void_p] * 7),
    ("phmm_batch_upload", C.c_int, [C.c_void_p, u8p, u8p, u8p, u8p, u8p, u8p]),
    ("phmm_batch_launch", C.c_int, [C.c_void_p, C.c_void_p]),
    ("phmm_batch_download", C.c_int, [C.c_void_p, f64p]),
    ("phmm_batch_status", C.c_int, [C.c_void_p]),
    ("phmm_batch_cells", C.c_uint64, [C.c_void_p]),
    ("phmm_batch_algorithmic_bytes", C.c_uint64, [C.c_void_p]),
    ("phmm_batch_num_launches", C.c_uint32, [C.c_void_p]),
    ("phmm_batch_executed_cells", C.c_uint64, [C.c_void_p]),
    ("phmm_batch_dominant_kernel", C.c_char_p, [C.c_void_p]),
    ("phmm_plan_describe", C.c_int, [C.c_uint, C.c_uint32, C.c_uint32, u32p, u32p, u32p, u32p, C.POINTER(PlanInfo)]),
    ("phmm_engine_compute", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u8p, u8p, u8p, u8p,
                                      u32p, u8p, C.POINTER(C.c_int32), u64p, f64p, u8p]),
    ("phmm_engine_submit", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u8p, u8p, u8p, u8p,
                                     u32p, u8p, C.POINTER(C.c_int32), u64p, f64p, u8p, C.POINTER(C.c_uint64)]),
    ("phmm_engine_compute_multi", C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u8p, u8p, u8p, u8p,
                                            u32p, u8p, C.POINTER(C.c_int32), u64p, f64p, u8p]),
    ("phmm_sw_align", C.c_int, [C.c_void_p, C.c_uint32, u32p, u8p, u32p, u8p, C.c_void_p, C.c_int, u64p, u32p, u32p,
                                C.POINTER(C.c_int32)]),
    ("phmm_sw_align_indexed", C.c_int, [C.c_void_p, C.c_uint32, u32p, u8p, C.c_uint32, u32p, u32p, u8p, C.c_void_p, C.c_int, u64p,
                                        u32p, u32p, C.POINTER(C.c_int32)]),
    ("phmm_best_alleles", C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p, u64p, f64p, u8p, C.POINTER(C.c_int32), C.c_double,
                                    C.POINTER(C.c_int32), f64p, f64p]),
    ("phmm_realign_to_best", C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u32p, u8p, u64p, f64p, u8p,
                                       C.POINTER(C.c_int32), C.c_double, C.c_void_p, C.c_int, u64p, u32p, u32p,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), f64p, f64p]),
    ("phmm_project_to_reference", C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u32p, u8p, C.POINTER(C.c_int32), u64p, u32p, u32p,
                                            u32p, C.POINTER(C.c_int32), u64p, u32p, u32p, C.POINTER(C.c_int32), u32p, u32p, u64p, u32p, u32p,
                                            C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("phmm_realign_reads", C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p, u32p, u8p, u32p, u8p, u64p, f64p, u8p, C.POINTER(C.c_int32), C.c_double,
                                     C.c_void_p, C.c_int, C.POINTER(C.c_int32), u64p, u32p, u32p, u32p, u32p, u32p, u64p, u32p, u32p,
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), f64p, f64p]),
    ("phmm_region_compute", C.c_int, _REGION_ARGS),
    ("phmm_region_submit", C.c_int, _REGION_ARGS + [C.POINTER(C.c_uint64)]),
    ("phmm_region_compute_multi", C.c_int, [C.POINTER(C.c_void_p), C.c_uint32] + _REGION_ARGS[1:]),
    ("phmm_calculate_cigar", C.c_int, [C.c_void_p, C.c_uint32, u32p, u8p, u32p, u8p, C.c_void_p, C.c_int, u64p, u32p, u32p, C.POINTER(C.c_int32)]),
    ("phmm_set_switch", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("phmm_get_stat", C.c_uint64, [C.c_void_p, C.c_char_p]),
    ("phmm_build_info", C.c_char_p, []),
    ("phmm_server_trace", C.c_uint32, [C.c_int, C.c_void_p, C.c_uint32]),
    ("phmm_table_eps", C.c_size_t, [C.POINTER(f64p)]),
    ("phmm_table_match_to_match", C.c_size_t, [C.POINTER(f64p)]),
]

_lib = None


def load():
    """Load libphmm.so and bind every symbol of include/phmm.h.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "lorikeet_amd: %s not found -- the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)." % LIB_PATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64; if libphmm.so pulled in the system
    # copy first, torch's later initialisation would see "No HIP GPUs".  Import torch first (when it is
    # installed) so both resolve to the same runtime.  Plain C / Rust callers link the system runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
