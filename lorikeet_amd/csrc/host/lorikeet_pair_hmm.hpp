// C++ host layer above the C ABI (include/phmm.h): the reference's PairHMM surface, same names,
// argument meaning and error behaviour, so that a test written against Lorikeet's Rust API reads the
// same here (tests/cpp/reference_tests.cpp).  Header-only; all arithmetic happens in libphmm.so on the
// GPU -- nothing in this file computes a likelihood, and there is no CPU fallback.
//
//   reference (Rust)                                                     here (C++, namespace lorikeet)
//   src/pair_hmm/pair_hmm.rs:63-125        PairHMM::initialize           PairHMM::initialize
//   src/pair_hmm/pair_hmm.rs:189-191       do_not_use_tristate_correction PairHMM::do_not_use_tristate_correction
//   src/pair_hmm/pair_hmm.rs:217-341       compute_log10_likelihoods      PairHMM::compute_log10_likelihoods
//   src/pair_hmm/pair_hmm.rs:345-375       compute_likelihoods            PairHMM::compute_likelihoods
//   src/pair_hmm/pair_hmm.rs:377-379       get_log_likelihood_array       PairHMM::get_log_likelihood_array
//   gkl::pairhmm::forward() closure (pair_hmm.rs:348-366)                 forward(hap, read, quals, ins, del, gcp)
//   ...likelihood_calculation_engine.rs:60-94   PCRErrorModel             PCRErrorModel
//   ...engine.rs:129-167, 195-242          PairHMMLikelihoodCalculationEngine::{new, compute_read_likelihoods}
//   ...engine.rs:632-652                   PairHMMInputScoreImputator     PairHMMInputScoreImputator
//   ...engine.rs:654-672                   AVXMode                        AVXMode (+ Hip)
//   src/model/allele_likelihoods.rs:47,79-125,324-345  AlleleLikelihoods  AlleleLikelihoods (layout contract)
//   src/haplotype/haplotype.rs:263-275     Haplotype (Eq/Hash by bases)   Haplotype
//   src/assembly/assembly_result_set.rs:35 AssemblyResultSet (ordered set of haplotypes)
//   src/reads/read_utils.rs:23,372-416     BI/BD tags or flat Q45         HmmRead::base_{insertion,deletion}_qualities
//   src/model/allele_likelihoods.rs:1043-1166  best_alleles_breaking_ties_main, BestAllele  AssemblyBasedCallerUtils::best_alleles_breaking_ties_main
//   src/assembly/assembly_based_caller_utils.rs:187-246  tie-breaking priorities, realign_reads_to_their_best_haplotype (arithmetic)
//   src/reads/alignment_utils.rs:40-165     AlignmentUtils::create_read_aligned_to_ref (position and CIGAR)
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/phmm.h"

namespace lorikeet {

using Bytes = std::vector<uint8_t>;
inline Bytes bytes(const std::string &s) { return Bytes(s.begin(), s.end()); }

// Rust panics / asserts on this path become exceptions carrying the same message.
struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

enum class AVXMode { AVX, Hip, None };
inline AVXMode detect_mode() { return phmm_device_count() > 0 ? AVXMode::Hip : AVXMode::None; }  // engine.rs:660-671

enum class PCRErrorModel { None = 0, Hostile = 1, Aggresive = 2, Conservative = 3 };  // (sic) engine.rs:61-70
inline PCRErrorModel pcr_error_model_from_arg(std::string s) {                          // engine.rs:79-93
    std::transform(s.begin(), s.end(), s.begin(), ::tolower);
    if (s == "none") return PCRErrorModel::None;
    if (s == "hostile") return PCRErrorModel::Hostile;
    if (s == "aggressive") return PCRErrorModel::Aggresive;
    if (s == "conservative") return PCRErrorModel::Conservative;
    throw Panic("Unknown PCR Error Model");
}

namespace MathUtils {
inline double log_to_log10(double ln) { return ln * std::log10(std::exp(1.0)); }  // math_utils.rs:120-122
}
namespace QualityUtils {
inline double qual_to_error_prob_log10(double qual) { return qual * -0.1; }  // quality_utils.rs:37-40
inline double qual_to_prob(uint8_t q) { return 1.0 - std::pow(10.0, q / -10.0); }
inline double qual_to_error_prob(uint8_t q) { return std::pow(10.0, q / -10.0); }
}  // namespace QualityUtils

struct Haplotype {
    Bytes bases_;
    bool is_ref = false;
    size_t cigar_elements = 1;  // elements of the haplotype -> reference CIGAR (haplotype_alignment_tiebreaking_priority)
    std::vector<uint32_t> cigar;  // the haplotype -> reference CIGAR, BAM-encoded (empty: one M over the bases)
    size_t alignment_start_hap_wrt_ref = 0;
    void set_cigar(const std::vector<uint32_t> &c) {
        cigar = c;
        cigar_elements = c.size();
    }
    void set_alignment_start_hap_wrt_ref(size_t s) { alignment_start_hap_wrt_ref = s; }
    Haplotype() = default;
    Haplotype(const Bytes &b, bool is_reference) : bases_(b), is_ref(is_reference) {}
    Haplotype(const std::string &b, bool is_reference) : bases_(bytes(b)), is_ref(is_reference) {}
    const Bytes &get_bases() const { return bases_; }
    size_t len() const { return bases_.size(); }
    bool operator==(const Haplotype &o) const { return bases_ == o.bases_; }  // identity by bases only
};

// What the path needs of a BirdToolRead.
struct HmmRead {
    std::string name = "read";
    Bytes bases, quals;
    std::optional<Bytes> ins_quals, del_quals;  // BI / BD tags
    uint8_t mapq = 60;
    static constexpr uint8_t DEFAULT_INSERTION_DELETION_QUAL = 45;  // read_utils.rs:23
    HmmRead() = default;
    HmmRead(const Bytes &b, const Bytes &q) : bases(b), quals(q) {
        if (b.size() != q.size()) throw Panic("Read bases and read quals aren't the same size");
    }
    size_t len() const { return bases.size(); }
    Bytes base_insertion_qualities() const { return ins_quals ? *ins_quals : Bytes(len(), DEFAULT_INSERTION_DELETION_QUAL); }
    Bytes base_deletion_qualities() const { return del_quals ? *del_quals : Bytes(len(), DEFAULT_INSERTION_DELETION_QUAL); }
    bool operator==(const HmmRead &o) const { return name == o.name && bases == o.bases && quals == o.quals; }
};

class PairHMMInputScoreImputator {  // engine.rs:632-652
    uint8_t constant_gcp;

public:
    explicit PairHMMInputScoreImputator(uint8_t gcp) : constant_gcp(gcp) {}
    Bytes ins_open_penalties(const HmmRead &r) const { return r.base_insertion_qualities(); }
    Bytes del_open_penalties(const HmmRead &r) const { return r.base_deletion_qualities(); }
    Bytes gap_continuation_penalties(const HmmRead &r) const { return Bytes(r.len(), constant_gcp); }
};

// Row-major [allele][read] matrix, the shape of ndarray::Array2<f64> in values_by_sample_index.
struct Matrix {
    size_t rows = 0, cols = 0;
    std::vector<double> v;
    Matrix() = default;
    Matrix(size_t r, size_t c, double fill = 0.0) : rows(r), cols(c), v(r * c, fill) {}
    double &operator()(size_t a, size_t r) { return v[a * cols + r]; }
    double operator()(size_t a, size_t r) const { return v[a * cols + r]; }
};

class AlleleLikelihoods {  // allele_likelihoods.rs:47,79-125,324-345
public:
    std::vector<Haplotype> alleles_;  // unique by bases, insertion ordered (IndexSet)
    std::vector<size_t> samples;
    std::map<size_t, std::vector<HmmRead>> evidence_by_sample_index, filtered_evidence_by_sample_index;
    std::vector<Matrix> values_by_sample_index;
    std::optional<size_t> reference_allele_index;

    AlleleLikelihoods(const std::vector<Haplotype> &alleles, const std::vector<size_t> &samples_,
                      const std::map<size_t, std::vector<HmmRead>> &evidence)
        : samples(samples_), evidence_by_sample_index(evidence) {
        for (const auto &a : alleles)
            if (std::find(alleles_.begin(), alleles_.end(), a) == alleles_.end()) alleles_.push_back(a);
        for (size_t i = 0; i < alleles_.size(); ++i)
            if (alleles_[i].is_ref && !reference_allele_index) reference_allele_index = i;
        for (size_t s = 0; s < samples.size(); ++s)
            values_by_sample_index.emplace_back(alleles_.size(), evidence_by_sample_index[s].size());
    }
    const std::vector<Haplotype> &alleles() const { return alleles_; }
    size_t number_of_alleles() const { return alleles_.size(); }
    size_t evidence_count() const {
        size_t n = 0;
        for (const auto &kv : evidence_by_sample_index) n += kv.second.size();
        return n;
    }
    Matrix &sample_matrix(size_t s) { return values_by_sample_index[s]; }
    size_t index_of_allele(const Haplotype &h) const {
        auto it = std::find(alleles_.begin(), alleles_.end(), h);
        if (it == alleles_.end()) throw Panic("Could not map new order to old order as new index was not present in new list");
        return (size_t)(it - alleles_.begin());
    }
};

class AssemblyResultSet {  // ordered, unique haplotypes (assembly_result_set.rs:35)
public:
    std::vector<Haplotype> haplotypes;
    AssemblyResultSet() = default;
    explicit AssemblyResultSet(const Haplotype &ref) { add_haplotype(ref); }
    bool add_haplotype(const Haplotype &h) {
        if (std::find(haplotypes.begin(), haplotypes.end(), h) != haplotypes.end()) return false;
        haplotypes.push_back(h);
        return true;
    }
};

namespace detail {

struct HandleDeleter {
    void operator()(phmm_handle *h) const { phmm_destroy(h); }
};
using Handle = std::unique_ptr<phmm_handle, HandleDeleter>;

inline Handle make_handle(int device, unsigned flags) {
    phmm_handle *h = phmm_create(device, flags);
    if (!h) throw Panic(std::string("Running in HIP mode but no HIP device is available: ") + phmm_last_error(nullptr));
    return Handle(h);
}

// One region of (reads, haplotypes) flattened into the SoA the ABI takes.
struct Flat {
    std::vector<uint32_t> region_read_off{0, 0}, region_hap_off{0, 0}, read_off{0}, hap_off{0};
    std::vector<uint64_t> out_off{0, 0};
    Bytes read_bases, base_q, ins_q, del_q, gcp, hap_bases;
    void add_read(const Bytes &b, const Bytes &q, const Bytes &i, const Bytes &d, const Bytes &g) {
        // the reference asserts these (pair_hmm.rs:425-440)
        if (q.size() != b.size()) throw Panic("Read bases and read quals aren't the same size");
        if (i.size() != b.size()) throw Panic("Read bases and insertion gcp aren't the same size");
        if (d.size() != b.size()) throw Panic("Read bases and deletion gcp aren't the same size");
        if (g.size() != b.size()) throw Panic("Read bases and overal GCP aren't the same size");
        read_bases.insert(read_bases.end(), b.begin(), b.end());
        base_q.insert(base_q.end(), q.begin(), q.end());
        ins_q.insert(ins_q.end(), i.begin(), i.end());
        del_q.insert(del_q.end(), d.begin(), d.end());
        gcp.insert(gcp.end(), g.begin(), g.end());
        read_off.push_back((uint32_t)read_bases.size());
        region_read_off[1] += 1;
    }
    void add_hap(const Bytes &h) {
        hap_bases.insert(hap_bases.end(), h.begin(), h.end());
        hap_off.push_back((uint32_t)hap_bases.size());
        region_hap_off[1] += 1;
    }
    size_t n_out() const { return (size_t)region_read_off[1] * region_hap_off[1]; }
};

// The reference builds a PairHMM per region and clones its engine per rayon task (assembly_region_walker.rs:227), so
// these objects are created and dropped all the time and used from many threads at once.  They therefore hold no device
// state: every object of a process shares one engine handle per (device, flags) and goes through phmm_submit /
// phmm_wait, the thread-safe pair of the ABI, which also computes the regions of all waiting threads as one batch.
inline phmm_handle *shared_handle(unsigned flags) {
    static Handle plain = make_handle(0, 0), no_tristate = make_handle(0, PHMM_FLAG_NO_TRISTATE);
    return (flags & PHMM_FLAG_NO_TRISTATE) ? no_tristate.get() : plain.get();
}

inline void check(phmm_handle *h, int rc) {
    if (rc == PHMM_OK) return;
    const std::string msg = phmm_last_error(h);
    // keep the reference's convention: violations panic (pair_hmm.rs:372, :478-481)
    throw Panic(msg.empty() ? "phmm error " + std::to_string(rc) : msg);
}

}  // namespace detail

// gkl::pairhmm::forward()'s closure: one (read, haplotype) log10 likelihood, tristate correction on.
inline double forward(const Bytes &hap, const Bytes &read, const Bytes &quals, const Bytes &ins, const Bytes &del,
                      const Bytes &gcp) {
    phmm_handle *h = detail::shared_handle(0);
    detail::Flat f;
    f.add_read(read, quals, ins, del, gcp);
    f.add_hap(hap);
    f.out_off[1] = 1;
    double out = 0.0;
    uint64_t ticket = 0;
    detail::check(h, phmm_submit(h, 1, f.region_read_off.data(), f.region_hap_off.data(), f.read_off.data(),
                                 f.read_bases.data(), f.base_q.data(), f.ins_q.data(), f.del_q.data(), f.gcp.data(),
                                 f.hap_off.data(), f.hap_bases.data(), f.out_off.data(), &out, &ticket));
    detail::check(h, phmm_wait(h, ticket));
    return out;
}

class PairHMM {
    bool initialized = false;
    bool no_tristate = false;
    std::vector<double> m_log_likelihood_array;
    std::vector<Bytes> m_haplotype_data_array;
    std::vector<Haplotype> haplotype_list;  // list index == position (haplotype_to_haplotype_list_index_map)
    AVXMode avx_mode = AVXMode::Hip;

    phmm_handle *engine() { return detail::shared_handle(no_tristate ? PHMM_FLAG_NO_TRISTATE : 0); }

public:
    // pair_hmm.rs:63-108 (the AVX arm's shape: haplotype byte slices + list-index map)
    static PairHMM initialize(const std::vector<Haplotype> &haplotypes,
                              const std::map<size_t, std::vector<HmmRead>> & /*per_sample_read_list*/, AVXMode avx_mode) {
        if (avx_mode != AVXMode::Hip) throw Panic("Running in AVX Mode but AVX is unavailable.");  // :372
        PairHMM p;
        for (const auto &h : haplotypes) {
            p.m_haplotype_data_array.push_back(h.get_bases());
            p.haplotype_list.push_back(h);
        }
        p.initialized = true;
        p.avx_mode = avx_mode;
        return p;
    }
    // quick_initialize(max_read_length, max_haplotype_length) (:128-165): sizes are irrelevant on the GPU
    static PairHMM quick_initialize(size_t, size_t) {
        PairHMM p;
        p.initialized = true;
        return p;
    }
    void do_not_use_tristate_correction() { no_tristate = true; }
    const std::vector<double> &get_log_likelihood_array() const { return m_log_likelihood_array; }

    // pair_hmm.rs:345-375: read-major x haplotype-list-order, one GPU batch
    void compute_likelihoods(detail::Flat &f) {
        f.out_off[1] = f.n_out();
        m_log_likelihood_array.assign(f.n_out(), 0.0);
        if (f.n_out() == 0) return;
        uint64_t ticket = 0;
        detail::check(engine(), phmm_submit(engine(), 1, f.region_read_off.data(), f.region_hap_off.data(), f.read_off.data(),
                                            f.read_bases.data(), f.base_q.data(), f.ins_q.data(), f.del_q.data(),
                                            f.gcp.data(), f.hap_off.data(), f.hap_bases.data(), f.out_off.data(),
                                            m_log_likelihood_array.data(), &ticket));
        detail::check(engine(), phmm_wait(engine(), ticket));
    }

    // pair_hmm.rs:217-267
    void compute_log10_likelihoods(size_t sample_index, AlleleLikelihoods &allele_likelihoods,
                                   const std::vector<HmmRead> &processed_reads,
                                   const PairHMMInputScoreImputator &input_score_imputator) {
        if (processed_reads.empty()) return;  // :224
        if (!initialized) throw Panic("Must call initialize before calling compute_read_likelihood_given_haplotype_log10");
        if (m_haplotype_data_array.empty())  // quick_initialize'd object: take the alleles, like the scalar arm (:272-337)
            for (const auto &a : allele_likelihoods.alleles()) {
                m_haplotype_data_array.push_back(a.get_bases());
                haplotype_list.push_back(a);
            }
        detail::Flat f;
        for (const auto &r : processed_reads)
            f.add_read(r.bases, r.quals, input_score_imputator.ins_open_penalties(r),
                       input_score_imputator.del_open_penalties(r), input_score_imputator.gap_continuation_penalties(r));
        for (const auto &h : m_haplotype_data_array) f.add_hap(h);
        compute_likelihoods(f);
        const size_t num_haplotypes = m_haplotype_data_array.size();
        Matrix &vals = allele_likelihoods.values_by_sample_index[sample_index];
        size_t read_index = 0;
        for (size_t r = 0; r < processed_reads.size(); ++r) {
            for (size_t a = 0; a < allele_likelihoods.number_of_alleles(); ++a) {
                // the order of haplotypes in the list and in the allele map may differ (:249-263)
                const Haplotype &al = allele_likelihoods.alleles()[a];
                size_t idx = num_haplotypes;
                for (size_t i = 0; i < haplotype_list.size(); ++i)
                    if (haplotype_list[i] == al) idx = i;  // later duplicates win, like HashMap::insert
                if (idx == num_haplotypes)
                    throw Panic("Could not map new order to old order as new index was not present in new list");
                vals(a, r) = m_log_likelihood_array[read_index + idx];
            }
            read_index += num_haplotypes;
        }
    }

    // pair_hmm.rs:405-501 (scalar entry point used by the reference's analytic tests); caching arguments are
    // accepted and ignored: every pair is recomputed in full, which the reference's own test pins as equal (:725-814)
    double compute_read_likelihood_given_haplotype_log10(const Bytes &haplotype_bases, const Bytes &read_bases,
                                                         const Bytes &read_quals, const Bytes &insertion_gop,
                                                         const Bytes &deletion_gop, const Bytes &overall_gcp,
                                                         bool /*recache_read_values*/,
                                                         const std::optional<Bytes> & /*next_haplotype_bases*/) {
        if (!initialized) throw Panic("Must call initialize before calling compute_read_likelihood_given_haplotype_log10");
        detail::Flat f;
        f.add_read(read_bases, read_quals, insertion_gop, deletion_gop, overall_gcp);
        f.add_hap(haplotype_bases);
        compute_likelihoods(f);
        return m_log_likelihood_array[0];
    }
};

class PairHMMLikelihoodCalculationEngine {
    phmm_engine_config cfg{};

public:
    // engine.rs:129-167, argument for argument
    PairHMMLikelihoodCalculationEngine(uint8_t constant_gcp, double log10_global_read_mismapping_rate,
                                       PCRErrorModel pcr_error_model, uint8_t base_quality_score_threshold,
                                       bool dynamic_read_disqualification, double read_disqualification_scale,
                                       double expected_error_rate_per_base,
                                       bool symmetrically_normalize_alleles_to_reference,
                                       bool disable_cap_read_qualities_to_mapq, bool modify_soft_clipped_bases,
                                       AVXMode avx_mode) {
        if (!modify_soft_clipped_bases)
            throw Panic("modify_soft_clipped_bases = false is not modelled (reads reach the path already hard-clipped)");
        if (avx_mode != AVXMode::Hip) throw Panic("Running in AVX Mode but AVX is unavailable.");
        cfg.constant_gcp = constant_gcp;
        cfg.pcr_error_model = (uint8_t)pcr_error_model;
        cfg.base_quality_score_threshold = base_quality_score_threshold;
        cfg.dynamic_read_disqualification = dynamic_read_disqualification;
        cfg.symmetrically_normalize_alleles_to_reference = symmetrically_normalize_alleles_to_reference;
        cfg.disable_cap_read_qualities_to_mapq = disable_cap_read_qualities_to_mapq;
        cfg.log10_global_read_mismapping_rate = log10_global_read_mismapping_rate;
        cfg.read_disqualification_scale = read_disqualification_scale;
        cfg.expected_error_rate_per_base = expected_error_rate_per_base;
        (void)detail::shared_handle(0);  // fail here, like the reference's mode check, if there is no device
    }

    // engine.rs:195-242
    AlleleLikelihoods compute_read_likelihoods(AssemblyResultSet &assembly_result_set, const std::vector<size_t> &samples,
                                               std::map<size_t, std::vector<HmmRead>> per_sample_read_list) {
        for (size_t i = 0; i < samples.size(); ++i) per_sample_read_list[i];  // :201-205
        const std::vector<Haplotype> &haplotypes = assembly_result_set.haplotypes;
        AlleleLikelihoods result(haplotypes, samples, per_sample_read_list);
        const size_t nh = result.number_of_alleles();
        // all samples of the region in one device call (every step is per read)
        std::vector<uint32_t> rro{0, 0}, rho{0, (uint32_t)nh}, ro{0}, ho{0};
        Bytes bases, quals, ins, del, mapq, haps;
        bool tags = false;
        for (size_t s = 0; s < samples.size(); ++s)
            for (const auto &r : result.evidence_by_sample_index[s]) tags |= r.ins_quals.has_value() || r.del_quals.has_value();
        for (size_t s = 0; s < samples.size(); ++s)
            for (const auto &r : result.evidence_by_sample_index[s]) {
                bases.insert(bases.end(), r.bases.begin(), r.bases.end());
                quals.insert(quals.end(), r.quals.begin(), r.quals.end());
                if (tags) {
                    const Bytes i = r.base_insertion_qualities(), d = r.base_deletion_qualities();
                    ins.insert(ins.end(), i.begin(), i.end());
                    del.insert(del.end(), d.begin(), d.end());
                }
                mapq.push_back(r.mapq);
                ro.push_back((uint32_t)bases.size());
                rro[1] += 1;
            }
        for (const auto &h : result.alleles()) {
            haps.insert(haps.end(), h.get_bases().begin(), h.get_bases().end());
            ho.push_back((uint32_t)haps.size());
        }
        const size_t nr = rro[1];
        std::vector<uint64_t> oo{0, (uint64_t)nr * nh};
        std::vector<double> out(nr * nh);
        Bytes keep(nr, 1);
        int32_t ref = result.reference_allele_index ? (int32_t)*result.reference_allele_index : -1;
        if (nr && nh) {
            phmm_handle *h = detail::shared_handle(0);
            uint64_t ticket = 0;
            detail::check(h, phmm_engine_submit(h, &cfg, 1, rro.data(), rho.data(), ro.data(), bases.data(), quals.data(),
                                                tags ? ins.data() : nullptr, tags ? del.data() : nullptr, mapq.data(),
                                                ho.data(), haps.data(), &ref, oo.data(), out.data(), keep.data(), &ticket));
            detail::check(h, phmm_wait(h, ticket));
        }
        // scatter [read][hap] -> [allele, read] per sample, applying the keep mask the way
        // remove_evidence_by_index does (allele_likelihoods.rs:968-1018): compact, NaN tail
        size_t pos = 0;
        for (size_t s = 0; s < samples.size(); ++s) {
            std::vector<HmmRead> &reads = result.evidence_by_sample_index[s];
            const size_t n = reads.size();
            Matrix m(nh, n, std::nan(""));
            std::vector<HmmRead> kept, removed;
            for (size_t r = 0; r < n; ++r) {
                if (keep[pos + r]) {
                    for (size_t a = 0; a < nh; ++a) m(a, kept.size()) = out[(pos + r) * nh + a];
                    kept.push_back(reads[r]);
                } else {
                    removed.push_back(reads[r]);
                }
            }
            pos += n;
            result.values_by_sample_index[s] = m;
            result.filtered_evidence_by_sample_index[s] = removed;
            reads = kept;
        }
        return result;
    }
};

}  // namespace lorikeet

// ---------------------------------------------------------------------------------------------------------------------
// Smith-Waterman (reference src/smith_waterman/smith_waterman_aligner.rs): same names, argument order and results;
// every alignment runs on the device through phmm_sw_align (there is no CPU path).
// ---------------------------------------------------------------------------------------------------------------------
namespace lorikeet {

struct Parameters {  // gkl::smithwaterman::Parameters::new(match_value, mismatch_penalty, gap_open_penalty, gap_extend_penalty)
    int32_t match_value, mismatch_penalty, gap_open_penalty, gap_extend_penalty;
};
enum class OverhangStrategy { SoftClip = PHMM_SW_SOFTCLIP, InDel = PHMM_SW_INDEL, LeadingInDel = PHMM_SW_LEADING_INDEL, Ignore = PHMM_SW_IGNORE };

// smith_waterman_aligner.rs:11-26
static const Parameters ORIGINAL_DEFAULT{3, -1, -4, -3};
static const Parameters STANDARD_NGS{25, -50, -110, -6};
static const Parameters NEW_SW_PARAMETERS{200, -150, -260, -11};
static const Parameters ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS{10, -15, -30, -5};

struct SmithWatermanAlignmentResult {  // :454-476
    std::vector<uint32_t> cigar;  // BAM encoding: (length << 4) | op, M = 0, I = 1, D = 2, S = 4
    int32_t alignment_offset = 0;
    int32_t get_alignment_offset() const { return alignment_offset; }
    std::string get_cigar() const {  // CigarString::to_string()
        static const char ops[] = "MIDNSHP=X";
        std::string out;
        for (uint32_t e : cigar) out += std::to_string(e >> 4) + ops[e & 15];
        return out;
    }
};

class SmithWatermanAligner {
public:
    // align(reference, alternate, parameters, overhang_strategy, avx_mode) (:47-107); AVXMode is accepted for source
    // compatibility, the device is the only arm here
    static SmithWatermanAlignmentResult align(const Bytes &reference, const Bytes &alternate, const Parameters &parameters,
                                              OverhangStrategy overhang_strategy, AVXMode = AVXMode::Hip) {
        return align_batch({{reference, alternate}}, parameters, overhang_strategy)[0];
    }

    // Many pairs under one parameter set and strategy: what each call site of the reference has in hand (reads ->
    // best haplotype, src/reads/alignment_utils.rs:40-70; haplotypes -> reference, src/reads/cigar_utils.rs:358-405).
    static std::vector<SmithWatermanAlignmentResult> align_batch(const std::vector<std::pair<Bytes, Bytes>> &pairs,
                                                                 const Parameters &parameters, OverhangStrategy strategy) {
        static std::mutex mu;  // phmm_sw_align is one-thread-per-handle like every entry point but submit / wait
        static detail::Handle handle = detail::make_handle(0, 0);
        std::vector<uint32_t> ref_off{0}, alt_off{0};
        Bytes ref, alt;
        for (const auto &pr : pairs) {
            if (pr.first.empty() || pr.second.empty())  // :65-68
                throw Panic("non-empty sequences are required for the Smith-Waterman calculation");
            ref.insert(ref.end(), pr.first.begin(), pr.first.end());
            alt.insert(alt.end(), pr.second.begin(), pr.second.end());
            ref_off.push_back((uint32_t)ref.size());
            alt_off.push_back((uint32_t)alt.size());
        }
        const uint32_t n = (uint32_t)pairs.size();
        std::vector<uint64_t> cap(n, 24), cig_off(n + 1, 0);
        std::vector<uint32_t> cigar, n_cig(n);
        std::vector<int32_t> off(n);
        const phmm_sw_parameters prm{parameters.match_value, parameters.mismatch_penalty, parameters.gap_open_penalty,
                                     parameters.gap_extend_penalty};
        std::lock_guard<std::mutex> lock(mu);
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (uint32_t a = 0; a < n; ++a) cig_off[a + 1] = cig_off[a] + cap[a];
            cigar.assign(cig_off[n], 0);
            const int rc = phmm_sw_align(handle.get(), n, ref_off.data(), ref.data(), alt_off.data(), alt.data(), &prm,
                                         (int)strategy, cig_off.data(), cigar.data(), n_cig.data(), off.data());
            if (rc == PHMM_ERR_CIGAR_CAPACITY && attempt == 0) {  // the library reports the sizes: once more with those
                for (uint32_t a = 0; a < n; ++a) cap[a] = std::max<uint64_t>(cap[a], n_cig[a]);
                continue;
            }
            detail::check(handle.get(), rc);
            break;
        }
        std::vector<SmithWatermanAlignmentResult> out(n);
        for (uint32_t a = 0; a < n; ++a) {
            out[a].cigar.assign(cigar.begin() + cig_off[a], cigar.begin() + cig_off[a] + n_cig[a]);
            out[a].alignment_offset = off[a];
        }
        return out;
    }
};


// ---------------------------------------------------------------------------------------------------------------------
// Best alleles and the realignment of reads to them (reference src/model/allele_likelihoods.rs:457-554, :1043-1166;
// src/assembly/assembly_based_caller_utils.rs:187-246): phmm_best_alleles / phmm_realign_to_best.
// ---------------------------------------------------------------------------------------------------------------------
struct BestAllele {  // allele_likelihoods.rs:1119-1166
    std::optional<size_t> allele_index;
    size_t sample_index = 0, evidence_index = 0;
    double likelihood = 0.0, confidence = 0.0;
    static constexpr double LOG_10_INFORMATIVE_THRESHOLD = 0.2;  // :17
    bool is_informative() const { return confidence > LOG_10_INFORMATIVE_THRESHOLD; }
};

struct AssemblyBasedCallerUtils {
    // :187-195 and :197-199
    static int32_t haplotype_alignment_tiebreaking_priority(const Haplotype &h) { return (h.is_ref ? 1 : 0) + 1 - (int32_t)h.cigar_elements; }
    static int32_t reference_tiebreaking_priority(const Haplotype &h) { return h.is_ref ? 1 : 0; }

    // AlleleLikelihoods::best_alleles_breaking_ties_main(tie_breaking_priority) (:1043-1095): every sample, every unit
    // of evidence.  With `alignments` the reads are also aligned to their best haplotype in the same call
    // (realign_reads_to_their_best_haplotype, :208-246: SoftClip, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS).
    template <class Priority>
    static std::vector<BestAllele> best_alleles_breaking_ties_main(const AlleleLikelihoods &lk, Priority tie_breaking_priority,
                                                                   std::vector<SmithWatermanAlignmentResult> *alignments = nullptr) {
        static std::mutex mu;
        static detail::Handle handle = detail::make_handle(0, 0);
        const size_t A = lk.alleles_.size(), S = lk.samples.size();
        std::vector<uint32_t> rro{0}, rho{0}, read_off{0}, hap_off{0};
        std::vector<uint64_t> oo{0};
        std::vector<double> values;  // per sample [read][allele]
        std::vector<int32_t> pri;
        Bytes reads, haps;
        for (size_t s = 0; s < S; ++s) {
            const Matrix &m = lk.values_by_sample_index[s];
            const auto ev = lk.evidence_by_sample_index.find(s);
            const size_t n = std::min(ev == lk.evidence_by_sample_index.end() ? 0 : ev->second.size(), m.cols);  // :1083-1089
            for (size_t r = 0; r < n; ++r) {
                for (size_t a = 0; a < A; ++a) values.push_back(m(a, r));
                const Bytes &b = ev->second[r].bases;
                reads.insert(reads.end(), b.begin(), b.end());
                read_off.push_back((uint32_t)reads.size());
            }
            for (const Haplotype &h : lk.alleles_) {  // every sample sees the same alleles
                pri.push_back(tie_breaking_priority(h));
                haps.insert(haps.end(), h.bases_.begin(), h.bases_.end());
                hap_off.push_back((uint32_t)haps.size());
            }
            rro.push_back(rro.back() + (uint32_t)n);
            rho.push_back(rho.back() + (uint32_t)A);
            oo.push_back(oo.back() + (uint64_t)n * A);
        }
        const uint32_t n_reads = rro.back();
        std::vector<int32_t> best(n_reads), off(n_reads);
        std::vector<double> like(n_reads), conf(n_reads);
        std::vector<uint64_t> cig_off(n_reads + 1, 0), cap(n_reads, 16);
        std::vector<uint32_t> cigar, n_cig(n_reads);
        const phmm_sw_parameters prm{ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.match_value, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.mismatch_penalty,
                                     ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.gap_open_penalty, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.gap_extend_penalty};
        {
            std::lock_guard<std::mutex> lock(mu);
            if (!alignments) {
                detail::check(handle.get(), phmm_best_alleles(handle.get(), (uint32_t)S, rro.data(), rho.data(), oo.data(), values.data(), nullptr,
                                                              pri.data(), BestAllele::LOG_10_INFORMATIVE_THRESHOLD, best.data(), like.data(), conf.data()));
            } else {
                for (int attempt = 0; attempt < 2; ++attempt) {
                    for (uint32_t a = 0; a < n_reads; ++a) cig_off[a + 1] = cig_off[a] + cap[a];
                    cigar.assign(cig_off[n_reads], 0);
                    const int rc = phmm_realign_to_best(handle.get(), (uint32_t)S, rro.data(), rho.data(), read_off.data(), reads.data(), hap_off.data(),
                                                        haps.data(), oo.data(), values.data(), nullptr, pri.data(), BestAllele::LOG_10_INFORMATIVE_THRESHOLD,
                                                        &prm, PHMM_SW_SOFTCLIP, cig_off.data(), cigar.data(), n_cig.data(), off.data(), best.data(),
                                                        like.data(), conf.data());
                    if (rc == PHMM_ERR_CIGAR_CAPACITY && attempt == 0) {
                        for (uint32_t a = 0; a < n_reads; ++a) cap[a] = std::max<uint64_t>(cap[a], n_cig[a]);
                        continue;
                    }
                    detail::check(handle.get(), rc);
                    break;
                }
            }
        }
        std::vector<BestAllele> out(n_reads);
        if (alignments) alignments->assign(n_reads, SmithWatermanAlignmentResult());
        for (size_t s = 0; s < S; ++s)
            for (uint32_t r = rro[s]; r < rro[s + 1]; ++r) {
                BestAllele &b = out[r];
                b.sample_index = s;
                b.evidence_index = r - rro[s];
                if (best[r] >= 0) b.allele_index = (size_t)best[r];
                b.likelihood = like[r];
                b.confidence = conf[r];
                if (alignments && best[r] >= 0) {
                    (*alignments)[r].cigar.assign(cigar.begin() + cig_off[r], cigar.begin() + cig_off[r] + n_cig[r]);
                    (*alignments)[r].alignment_offset = off[r];
                }
            }
        return out;
    }
};


// ---------------------------------------------------------------------------------------------------------------------
// AlignmentUtils::create_read_aligned_to_ref (reference src/reads/alignment_utils.rs:40-165): the read aligned to the
// haplotype (phmm_sw_align_indexed) and that alignment projected onto the reference (phmm_project_to_reference).
// ---------------------------------------------------------------------------------------------------------------------
inline std::vector<uint32_t> parse_cigar(const std::string &text) {  // CigarString::try_from
    static const std::string ops = "MIDNSHP=X";
    std::vector<uint32_t> out;
    uint32_t n = 0;
    for (char ch : text) {
        if (ch >= '0' && ch <= '9') {
            n = n * 10 + (uint32_t)(ch - '0');
        } else {
            const size_t op = ops.find(ch);
            if (op == std::string::npos) throw Panic("bad CIGAR operator");
            out.push_back((n << 4) | (uint32_t)op);
            n = 0;
        }
    }
    return out;
}
inline std::string cigar_to_string(const std::vector<uint32_t> &c) {
    SmithWatermanAlignmentResult r;
    r.cigar = c;
    return r.get_cigar();
}

struct AlignedRead {  // what create_read_aligned_to_ref changes of the read: pos() and cigar(); `realigned` false = the clone
    bool realigned = false;
    int64_t pos = 0;
    std::vector<uint32_t> cigar;
};

struct AlignmentUtils {
    // original_read: the read minus its soft clips in `bases` (ReadClipper::hard_clip_soft_clipped_bases, :47-50) and its
    // CIGAR before realignment in `original_cigar` (only the clips are used, :135-143)
    static AlignedRead create_read_aligned_to_ref(const Bytes &read_minus_soft_clips, const std::vector<uint32_t> &original_cigar,
                                                  const Haplotype &haplotype, const Haplotype &ref_haplotype, size_t reference_start) {
        static std::mutex mu;
        static detail::Handle handle = detail::make_handle(0, 0);
        if (read_minus_soft_clips.empty() || haplotype.bases_.empty() || ref_haplotype.bases_.empty())
            throw Panic("non-empty sequences are required for the Smith-Waterman calculation");
        // one region: [reference haplotype, haplotype], one read
        Bytes haps = ref_haplotype.bases_;
        haps.insert(haps.end(), haplotype.bases_.begin(), haplotype.bases_.end());
        const uint32_t hap_off[3] = {0, (uint32_t)ref_haplotype.bases_.size(), (uint32_t)haps.size()};
        const uint32_t read_off[2] = {0, (uint32_t)read_minus_soft_clips.size()}, rro[2] = {0, 1}, rho[2] = {0, 2}, ref_index[1] = {1};
        const phmm_sw_parameters prm{ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.match_value, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.mismatch_penalty,
                                     ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.gap_open_penalty, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS.gap_extend_penalty};
        auto whole = [](const Haplotype &h) { return h.cigar.empty() ? std::vector<uint32_t>{(uint32_t)h.bases_.size() << 4} : h.cigar; };
        const std::vector<uint32_t> c_ref = whole(ref_haplotype), c_hap = whole(haplotype);
        std::vector<uint32_t> hap_cigar = c_ref;
        hap_cigar.insert(hap_cigar.end(), c_hap.begin(), c_hap.end());
        const uint32_t hap_cigar_off[3] = {0, (uint32_t)c_ref.size(), (uint32_t)hap_cigar.size()};
        const uint32_t hap_start[2] = {(uint32_t)ref_haplotype.alignment_start_hap_wrt_ref, (uint32_t)haplotype.alignment_start_hap_wrt_ref};
        const uint32_t oc_off[2] = {0, (uint32_t)original_cigar.size()};
        const int32_t region_ref_hap[1] = {0}, best[1] = {1};
        const uint64_t ref_start[1] = {(uint64_t)reference_start};
        const size_t cap = read_minus_soft_clips.size() + haplotype.bases_.size() + original_cigar.size() + hap_cigar.size() + 8;
        const uint64_t cig_off[2] = {0, cap};
        std::vector<uint32_t> sw(cap), out(cap);
        uint32_t n_sw = 0, n_out = 0;
        int32_t offset = 0, status = 0;
        int64_t pos = 0;
        std::lock_guard<std::mutex> lock(mu);
        detail::check(handle.get(), phmm_sw_align_indexed(handle.get(), 2, hap_off, haps.data(), 1, ref_index, read_off, read_minus_soft_clips.data(),
                                                          &prm, PHMM_SW_SOFTCLIP, cig_off, sw.data(), &n_sw, &offset));
        detail::check(handle.get(), phmm_project_to_reference(handle.get(), 1, rro, rho, read_off, read_minus_soft_clips.data(), hap_off, haps.data(),
                                                              region_ref_hap, ref_start, hap_cigar_off, hap_cigar.data(), hap_start, best, cig_off,
                                                              sw.data(), &n_sw, &offset, oc_off, original_cigar.data(), cig_off, out.data(), &n_out,
                                                              &pos, &status));
        if (status < 0) throw Panic("create_read_aligned_to_ref: the reference panics on this read (status " + std::to_string(status) + ")");
        AlignedRead r;
        r.realigned = status == PHMM_PROJECT_REALIGNED;
        r.pos = pos;
        r.cigar.assign(out.begin(), out.begin() + n_out);
        return r;
    }
};


struct CigarUtils {
    // calculate_cigar(ref_seq, alt_seq, strategy, sw_parameters, avx_mode) (src/reads/cigar_utils.rs:358-457): None -> nullopt
    static std::optional<std::vector<uint32_t>> calculate_cigar(const Bytes &ref_seq, const Bytes &alt_seq, OverhangStrategy strategy,
                                                                const Parameters &sw_parameters, AVXMode = AVXMode::Hip) {
        static std::mutex mu;
        static detail::Handle handle = detail::make_handle(0, 0);
        const uint32_t ref_off[2] = {0, (uint32_t)ref_seq.size()}, alt_off[2] = {0, (uint32_t)alt_seq.size()};
        const phmm_sw_parameters prm{sw_parameters.match_value, sw_parameters.mismatch_penalty, sw_parameters.gap_open_penalty,
                                     sw_parameters.gap_extend_penalty};
        const uint64_t cig_off[2] = {0, ref_seq.size() + alt_seq.size() + 4};
        std::vector<uint32_t> cigar(cig_off[1]);
        uint32_t n_cig = 0;
        int32_t status = 0;
        std::lock_guard<std::mutex> lock(mu);
        detail::check(handle.get(), phmm_calculate_cigar(handle.get(), 1, ref_off, ref_seq.data(), alt_off, alt_seq.data(), &prm, (int)strategy,
                                                         cig_off, cigar.data(), &n_cig, &status));
        if (status < 0) throw Panic("calculate_cigar: the reference panics on this pair (status " + std::to_string(status) + ")");
        if (status != 0) return std::nullopt;
        cigar.resize(n_cig);
        return cigar;
    }
};

}  // namespace lorikeet
