"""Host-side mirror of `PairHMMLikelihoodCalculationEngine` backed by `phmm_engine_compute`.

reference src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs
    PairHMMLikelihoodCalculationEngine::new             :129-167  -> PairHMMLikelihoodCalculationEngine(...)
    compute_read_likelihoods                            :195-242  -> .compute_read_likelihoods(...)
    PCRErrorModel                                       :60-94    -> PCRErrorModel
    AVXMode                                             :654-672  -> AVXMode (Hip is the only backend here)
reference src/assembly/assembly_result_set.rs:35 (ordered, unique haplotypes)  -> AssemblyResultSet

All numeric work (quality modification, PairHMM, normalisation, disqualification decision) runs on the
GPU in one `phmm_engine_compute` call; this file only marshals and applies the keep mask the way
AlleleLikelihoods::remove_evidence_by_index does (src/model/allele_likelihoods.rs:968-1018).
"""
import ctypes as C
import enum
import math

import numpy as np

from . import _lib
from .engine import HipPairHMMEngine, PhmmError
from .pair_hmm import AlleleLikelihoods, Haplotype, HmmRead  # noqa: F401


class PCRErrorModel(enum.IntEnum):
    NONE = 0
    HOSTILE = 1
    AGGRESSIVE = 2
    CONSERVATIVE = 3

    @staticmethod
    def from_arg(s):
        try:
            return {"none": PCRErrorModel.NONE, "hostile": PCRErrorModel.HOSTILE, "aggressive": PCRErrorModel.AGGRESSIVE,
                    "conservative": PCRErrorModel.CONSERVATIVE}[s.lower()]
        except KeyError:
            raise ValueError("Unknown PCR Error Model")  # engine.rs:89


class AVXMode(enum.Enum):
    Hip = "hip"

    @staticmethod
    def detect_mode():
        return AVXMode.Hip


def log_to_log10(ln):  # MathUtils::log_to_log10 (math_utils.rs:120-122)
    return ln * math.log10(math.e)


def qual_to_error_prob_log10(q):  # QualityUtils::qual_to_error_prob_log10 (quality_utils.rs:37-40)
    return q * -0.1


class AssemblyResultSet:
    """Ordered set of haplotypes, unique by bases (assembly_result_set.rs:35, haplotype.rs:263-275)."""

    def __init__(self, ref_haplotype=None):
        self.haplotypes = []
        if ref_haplotype is not None:
            self.add_haplotype(ref_haplotype)

    def add_haplotype(self, h):
        if h not in self.haplotypes:
            self.haplotypes.append(h)
            return True
        return False


class PairHMMLikelihoodCalculationEngine:
    def __init__(self, constant_gcp, log10_global_read_mismapping_rate, pcr_error_model, base_quality_score_threshold,
                 dynamic_read_disqualification, read_disqualification_scale, expected_error_rate_per_base,
                 symmetrically_normalize_alleles_to_reference, disable_cap_read_qualities_to_mapq,
                 modify_soft_clipped_bases=True, avx_mode=AVXMode.Hip, device_id=0, shared_engine=None):
        """shared_engine: a HipPairHMMEngine that several of these objects (one per worker thread, as the reference
        clones its engine per task, assembly_region_walker.rs:227) share; calls then go through phmm_engine_submit /
        phmm_wait, which computes the regions of all waiting workers as one batch."""
        if not modify_soft_clipped_bases:
            # The other branch of modify_read_qualities (:390-422) only differs for reads that still carry soft
            # clips, which the assembler has already removed upstream (SURVEY.md 8a trap 8/10).
            raise NotImplementedError("modify_soft_clipped_bases = false is not modelled (see SURVEY.md 8a)")
        self.cfg = _lib.EngineConfig()
        self.cfg.constant_gcp = int(constant_gcp)
        self.cfg.pcr_error_model = int(pcr_error_model)
        self.cfg.base_quality_score_threshold = int(base_quality_score_threshold)
        self.cfg.dynamic_read_disqualification = int(bool(dynamic_read_disqualification))
        self.cfg.symmetrically_normalize_alleles_to_reference = int(bool(symmetrically_normalize_alleles_to_reference))
        self.cfg.disable_cap_read_qualities_to_mapq = int(bool(disable_cap_read_qualities_to_mapq))
        self.cfg.log10_global_read_mismapping_rate = float(log10_global_read_mismapping_rate)
        self.cfg.read_disqualification_scale = float(read_disqualification_scale)
        self.cfg.expected_error_rate_per_base = float(expected_error_rate_per_base)
        self._shared = shared_engine is not None
        self._engine = shared_engine if self._shared else HipPairHMMEngine(device_id)

    # ---- low level: many regions, arrays in / arrays out -------------------------------------------------
    def compute_regions(self, regions):
        """regions: list of (reads: list[HmmRead], haplotypes: list[Haplotype]).
        Returns per region (normalised [read][hap] float64 matrix, keep bool[read])."""
        rro, rho, ro, ho, oo = [0], [0], [0], [0], [0]
        bases, quals, ins, dele, mapq, haps, ref = [], [], [], [], [], [], []
        any_tags = any(r.ins_quals is not None or r.del_quals is not None for reads, _ in regions for r in reads)
        for reads, hs in regions:
            for r in reads:
                bases.append(r.bases); quals.append(r.quals); mapq.append(r.mapq)
                if any_tags:
                    ins.append(r.base_insertion_qualities()); dele.append(r.base_deletion_qualities())
                ro.append(ro[-1] + len(r))
            ref_idx = -1
            for j, h in enumerate(hs):
                hb = np.frombuffer(h.get_bases(), np.uint8)
                haps.append(hb)
                ho.append(ho[-1] + len(hb))
                if h.is_ref and ref_idx < 0:
                    ref_idx = j
            ref.append(ref_idx)
            rro.append(rro[-1] + len(reads)); rho.append(rho[-1] + len(hs)); oo.append(oo[-1] + len(reads) * len(hs))

        def cat(xs):
            return np.ascontiguousarray(np.concatenate(xs), np.uint8) if xs else np.zeros(0, np.uint8)
        a = dict(rro=np.asarray(rro, np.uint32), rho=np.asarray(rho, np.uint32), ro=np.asarray(ro, np.uint32),
                 ho=np.asarray(ho, np.uint32), oo=np.asarray(oo, np.uint64), bases=cat(bases), quals=cat(quals),
                 mapq=np.asarray(mapq, np.uint8), haps=cat(haps), ref=np.asarray(ref, np.int32))
        out = np.empty(int(oo[-1]), np.float64)
        keep = np.zeros(len(mapq), np.uint8)
        p = lambda x, t: x.ctypes.data_as(t)  # noqa: E731
        ins_p = del_p = None  # NULL -> the reference's flat Q45 default (read_utils.rs:23)
        if any_tags:
            a["ins"], a["dele"] = cat(ins), cat(dele)
            ins_p, del_p = p(a["ins"], _lib.u8p), p(a["dele"], _lib.u8p)
        eng = self._engine
        args = (eng._h, C.byref(self.cfg), len(regions), p(a["rro"], _lib.u32p), p(a["rho"], _lib.u32p), p(a["ro"], _lib.u32p),
                p(a["bases"], _lib.u8p), p(a["quals"], _lib.u8p), ins_p, del_p, p(a["mapq"], _lib.u8p), p(a["ho"], _lib.u32p),
                p(a["haps"], _lib.u8p), p(a["ref"], C.POINTER(C.c_int32)), p(a["oo"], _lib.u64p), p(out, _lib.f64p),
                p(keep, _lib.u8p))
        if self._shared:
            ticket = C.c_uint64(0)
            code = eng.lib.phmm_engine_submit(*args, C.byref(ticket))
            if code == _lib.PHMM_OK:
                code = eng.lib.phmm_wait(eng._h, ticket.value)
        else:
            code = eng.lib.phmm_engine_compute(*args)
        if code != _lib.PHMM_OK:
            raise PhmmError(code, eng.last_error())
        res = []
        for g, (reads, hs) in enumerate(regions):
            m = out[oo[g]:oo[g + 1]].reshape(len(reads), len(hs)) if len(hs) else np.zeros((len(reads), 0))
            res.append((m, keep[rro[g]:rro[g + 1]].astype(bool)))
        return res

    # ---- the reference's surface ------------------------------------------------------------------------
    def compute_read_likelihoods(self, assembly_result_set, samples, per_sample_read_list):
        """engine.rs:195-242.  Returns AlleleLikelihoods: per sample an [allele, read] matrix of normalised
        log10 likelihoods, poorly modelled evidence moved to filtered_evidence_by_sample_index, matrix columns
        compacted and the vacated tail set to NaN (allele_likelihoods.rs:968-1018)."""
        per_sample_read_list = dict(per_sample_read_list)
        for i in range(len(samples)):
            per_sample_read_list.setdefault(i, [])
        haplotypes = list(assembly_result_set.haplotypes)
        result = AlleleLikelihoods(haplotypes, samples, per_sample_read_list)
        result.filtered_evidence_by_sample_index = {}
        order = [(s, r) for s in range(len(samples)) for r in result.evidence_by_sample_index[s]]
        if order and haplotypes:
            (m, keep), = self.compute_regions([([r for _, r in order], haplotypes)])
        else:
            m, keep = np.zeros((len(order), len(haplotypes))), np.ones(len(order), bool)
        pos = 0
        for s in range(len(samples)):
            reads = result.evidence_by_sample_index[s]
            n = len(reads)
            ms, ks = m[pos:pos + n], keep[pos:pos + n]
            pos += n
            vals = np.full((len(haplotypes), n), np.nan)
            kept = int(ks.sum())
            vals[:, :kept] = ms[ks].T
            result.values_by_sample_index[s] = vals
            result.filtered_evidence_by_sample_index[s] = [r for r, k in zip(reads, ks) if not k]
            result.evidence_by_sample_index[s] = [r for r, k in zip(reads, ks) if k]
        return result
