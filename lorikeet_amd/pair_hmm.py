"""Host-side mirror of the reference's PairHMM surface, backed ONLY by the HIP engine.

Names, argument meaning and error behaviour follow the reference so that the parity tests read
like the reference's own tests:

  reference src/pair_hmm/pair_hmm.rs
    PairHMM::initialize                :63-125   -> PairHMM.initialize
    PairHMM::compute_log10_likelihoods :217-341  -> PairHMM.compute_log10_likelihoods
    PairHMM::get_log_likelihood_array  :377-379  -> PairHMM.get_log_likelihood_array
    PairHMM::do_not_use_tristate_correction :189 -> PairHMM.do_not_use_tristate_correction
  gkl::pairhmm::forward()(hap, read, quals, ins, del, gcp) -> f64  (pair_hmm.rs:348-366)
                                                  -> forward(...)
  src/model/allele_likelihoods.rs:47,324-345      -> AlleleLikelihoods (layout contract only)
  PairHMMInputScoreImputator  (...engine.rs:632-652) -> PairHMMInputScoreImputator

(The Rust shim that binds the same C ABI from inside Lorikeet is in INTEGRATION.md.)
"""
import numpy as np

from .batch import Read, RegionBatch
from .engine import HipPairHMMEngine

DEFAULT_INDEL_QUAL = 45  # ReadUtils default BI/BD quality (src/reads/read_utils.rs:23,372-416)


class Haplotype:
    """Bases + is_ref; identity (eq/hash) is by bases only (src/haplotype/haplotype.rs:263-275)."""

    def __init__(self, bases, is_ref=False):
        self.bases = bytes(bases)
        self.is_ref = is_ref

    def get_bases(self):
        return self.bases

    def __len__(self):
        return len(self.bases)

    def __eq__(self, o):
        return isinstance(o, Haplotype) and self.bases == o.bases

    def __hash__(self):
        return hash(self.bases)


class HmmRead:
    """What the path needs of a BirdToolRead: bases, base quals, optional BI/BD tags, MAPQ."""

    def __init__(self, bases, quals, ins_quals=None, del_quals=None, mapq=60, name=b"read"):
        self.bases = np.frombuffer(bytes(bases), dtype=np.uint8).copy() if isinstance(bases, (bytes, bytearray, str)) \
            else np.array(bases, dtype=np.uint8)
        self.quals = np.array(quals, dtype=np.uint8)
        self.ins_quals = None if ins_quals is None else np.array(ins_quals, dtype=np.uint8)
        self.del_quals = None if del_quals is None else np.array(del_quals, dtype=np.uint8)
        self.mapq = mapq
        self.name = name

    def __len__(self):
        return len(self.bases)

    # ReadUtils::get_base_{insertion,deletion}_qualities: tag or flat Q45 (read_utils.rs:372-416)
    def base_insertion_qualities(self):
        return self.ins_quals if self.ins_quals is not None else np.full(len(self), DEFAULT_INDEL_QUAL, np.uint8)

    def base_deletion_qualities(self):
        return self.del_quals if self.del_quals is not None else np.full(len(self), DEFAULT_INDEL_QUAL, np.uint8)


class PairHMMInputScoreImputator:
    """pair_hmm_likelihood_calculation_engine.rs:632-652"""

    def __init__(self, gcp):
        self.constant_gcp = int(gcp)

    def ins_open_penalties(self, read):
        return read.base_insertion_qualities()

    def del_open_penalties(self, read):
        return read.base_deletion_qualities()

    def gap_continuation_penalties(self, read):
        return np.full(len(read), self.constant_gcp, np.uint8)


class AlleleLikelihoods:
    """Layout contract of the reference container: per sample an [allele, read] f64 matrix
    (allele_likelihoods.rs:47,324-345).  Alleles are unique by bases, insertion ordered."""

    def __init__(self, alleles, samples, evidence_by_sample_index):
        uniq = []
        for a in alleles:
            if a not in uniq:
                uniq.append(a)
        self.alleles = uniq
        self.samples = list(samples)
        self.evidence_by_sample_index = {s: list(evidence_by_sample_index.get(s, [])) for s in range(len(samples))}
        self.values_by_sample_index = [np.zeros((len(uniq), len(self.evidence_by_sample_index[s])), np.float64)
                                       for s in range(len(samples))]

    def number_of_alleles(self):
        return len(self.alleles)

    def sample_matrix(self, s):
        return self.values_by_sample_index[s]

    def evidence_count(self):
        return sum(len(v) for v in self.evidence_by_sample_index.values())


_ENGINES = {}


def _engine(device_id, no_tristate):
    key = (device_id, bool(no_tristate))
    if key not in _ENGINES:
        _ENGINES[key] = HipPairHMMEngine(device_id, do_not_use_tristate_correction=no_tristate)
    return _ENGINES[key]


def forward(hap_bases, read_bases, read_quals, ins_gop, del_gop, gcp, device_id=0, tristate=True):
    """One (read, haplotype) log10 likelihood -- the shape of gkl::pairhmm::forward()'s closure."""
    batch = RegionBatch.from_regions([([Read(read_bases, read_quals, ins_gop, del_gop, gcp)], [hap_bases])])
    return float(_engine(device_id, not tristate).compute(batch)[0])


class PairHMM:
    def __init__(self, haplotypes, device_id=0):
        self.m_haplotype_data_array = [h.get_bases() for h in haplotypes]
        self.haplotype_to_haplotype_list_index_map = {}
        for i, h in enumerate(haplotypes):
            self.haplotype_to_haplotype_list_index_map[h] = i  # later duplicates win, like HashMap::insert
        self.m_log_likelihood_array = []
        self._no_tristate = False
        self._device_id = device_id

    @staticmethod
    def initialize(haplotypes, per_sample_read_list=None, device_id=0):
        """pair_hmm.rs:63-108 (AVX arm shape: list of haplotype byte slices + list-index map)."""
        return PairHMM(haplotypes, device_id)

    def do_not_use_tristate_correction(self):
        self._no_tristate = True

    def get_log_likelihood_array(self):
        return self.m_log_likelihood_array

    def compute_likelihoods(self, read_data_array):
        """pair_hmm.rs:345-375: read-major x haplotype-list-order f64, one HIP batch."""
        batch = RegionBatch.from_regions([(read_data_array, self.m_haplotype_data_array)])
        self.m_log_likelihood_array = list(_engine(self._device_id, self._no_tristate).compute(batch))

    def compute_log10_likelihoods(self, sample_index, allele_likelihoods, processed_reads, input_score_imputator):
        """pair_hmm.rs:217-267: marshal ReadDataHolders, compute, scatter to [allele, read]."""
        if len(processed_reads) == 0:
            return  # :224
        num_haplotypes = allele_likelihoods.number_of_alleles()
        rda = [Read(r.bases, r.quals, input_score_imputator.ins_open_penalties(r),
                    input_score_imputator.del_open_penalties(r), input_score_imputator.gap_continuation_penalties(r))
               for r in processed_reads]
        self.compute_likelihoods(rda)
        n_list = len(self.m_haplotype_data_array)
        vals = allele_likelihoods.values_by_sample_index[sample_index]
        read_index = 0
        for r in range(len(processed_reads)):
            for a in range(num_haplotypes):
                idx = self.haplotype_to_haplotype_list_index_map[allele_likelihoods.alleles[a]]
                vals[a, r] = self.m_log_likelihood_array[read_index + idx]
            read_index += n_list
