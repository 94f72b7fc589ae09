// Internal declarations shared by the C ABI (phmm_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace phmm {

constexpr int WAVE = 64;              // CDNA4 wavefront
constexpr int MAX_WAVES_PER_BLOCK = 4;  // independent waves; a block is only a launch granule
constexpr int KMAX = 32;              // max haplotype columns per lane
constexpr int LDS_ROW_BYTES = 72;         // sizeof(RowConst): per read row staged in LDS

// Everything a forward launch needs.  All pointers are device pointers.
struct ForwardParams {
    // work list: items of this shape class.  item i -> global read index class_reads[i] (or i when null)
    const uint32_t *class_reads;
    uint32_t n_items;
    // batch metadata (device copies of the ABI offset arrays)
    const uint32_t *read_region;      // [n_reads] region id of each read
    const uint32_t *region_read_off;  // [n_regions+1]
    const uint32_t *region_hap_off;   // [n_regions+1]
    const uint32_t *read_off;         // [n_reads+1]
    const uint32_t *hap_off;          // [n_haps+1]
    const uint64_t *out_off;          // [n_regions+1]
    // payload
    const uint8_t *read_bases, *base_q, *ins_q, *del_q, *gcp, *hap_bases;
    double *out;
    // tables
    const double *eps;    // [256]  10^(-q/10)
    const double *eps_mis;  // [256]  mismatch prior: eps/3 (tristate) or eps
    const double *mm;     // triangular [256*257/2] match->match
    const double *ratio_mis;  // [256]  eps_mis[q] / (1 - eps[q]): mismatch/match prior ratio of pre-scaled rows (0 at q = 0)
    const double *inv_om;     // [256]  1 / (1 - eps[q]) (0 at q = 0)
    // scalars
    double initial_condition;        // 2^1020
    double initial_condition_log10;  // log10(2^1020), host libm
    uint32_t lds_rows;               // rows of LDS staging reserved per wave (>= longest read of the class)
    const uint8_t *redo;             // not null: only reads with redo[r] != 0 are computed (f64 pass of the f32-first mode)
    uint32_t pad_was_priority;       // (a wave-priority A/B of round 4 lived here: the time only moved to the other kernel)
    uint32_t cnd_select;             // 1: v_cndmask prior select (launches with < 2 waves per SIMD, K <= PHMM_CND_MAX_K)
    uint32_t *status;                // device status word (STATUS_POSITIVE | STATUS_RESCUE)
};

// Status word bits the forward kernels raise.
constexpr uint32_t STATUS_POSITIVE = 1u;  // some log10 likelihood came out > 0 (or NaN): pair_hmm.rs:478-481 asserts
constexpr uint32_t STATUS_RESCUE = 2u;    // some result lies below kRescueBelow: phmm_rescue recomputes those pairs
constexpr uint32_t STATUS_POSITIVE_FINAL = 4u;  // raised by phmm_rescue where a value is > 0 (or NaN) AFTER its pass: once that pass
                                                // has run (STATUS_RESCUE), this bit is the verdict and STATUS_POSITIVE -- which a fast
                                                // kernel may have raised for a pair the pass then replaced -- is not
__host__ __device__ inline bool status_positive(uint32_t bits, bool rescue_rides_inline) {
    return rescue_rides_inline && (bits & STATUS_RESCUE) ? (bits & STATUS_POSITIVE_FINAL) != 0 : (bits & STATUS_POSITIVE) != 0;
}
// Below this log10 likelihood the scaled row sum of the reference (2^1020 / H scale, pair_hmm.rs:515-529,598-614) gets
// close to the denormal range, where every rounding counts: the fast kernels (folded row constants, FMA contraction,
// the chained kernel's common 2^1010 start) are no longer within 1e-9 of the reference there, and they reach -inf at a
// different point.  Every pair whose result comes out below it (or -inf / NaN) is recomputed by phmm_rescue in the
// reference's own operation order, so that the underflow band agrees with the scalar arm including where it turns to
// -inf.  The fast kernels keep 1e-13 down to about -617 (scaled sum >= 2^-1040); -600 leaves 17 decades of margin.
constexpr double kRescueBelow = -600.0;
__host__ __device__ inline uint32_t status_bits(double v) {
    return (!(v <= 0.0) ? STATUS_POSITIVE : 0u) | (!(v >= kRescueBelow) ? STATUS_RESCUE : 0u);
}

// Launch the <L,K> instantiation.  Returns hipErrorInvalidValue if (L,K) is not instantiated.
hipError_t launch_forward(int L, int K, const ForwardParams &p, dim3 grid, int waves_per_block, size_t lds_bytes,
                          hipStream_t stream);
// Generic any-shape fallback (one thread per pair, rolling rows in global scratch).
struct GenericParams {
    ForwardParams f;
    double *scratch;        // [n_threads_total * 6 * (max_h+1)]
    uint32_t max_h;         // longest haplotype of the class
    const uint64_t *pair_first;  // [n_items+1] prefix of pairs per item-read (Nh of its region)
    uint64_t n_pairs;
    uint32_t n_blocks;      // grid the scratch was sized for (256 threads per block)
};
hipError_t launch_generic(const GenericParams &p, hipStream_t stream);

// Exact recomputation of the pairs below kRescueBelow (phmm_exact_kernels.hip): one thread per read scans its row of
// results and redoes what it finds, two rolling rows of M/I/D per thread in `scratch`.
struct RescueParams {
    ForwardParams f;
    uint32_t n_reads;    // reads of the batch (f.read_region etc. cover all of them)
    double *scratch;     // [n_threads * 6 * (max_h + 1)]
    uint32_t max_h;      // longest haplotype of the batch
    uint32_t n_blocks;   // grid the scratch was sized for (64 threads per block)
    uint32_t force;      // 0: return at once unless the status word carries STATUS_RESCUE (in-stream use)
};
hipError_t launch_rescue(const RescueParams &p, hipStream_t stream);
// threads (a multiple of 64) and bytes of scratch for a batch whose longest haplotype has max_h columns
inline void rescue_geometry(uint32_t max_h, uint32_t *n_blocks, size_t *scratch_bytes) {
    const size_t per_thread = 6ull * ((size_t)max_h + 1) * sizeof(double);
    size_t threads = (64ull << 20) / per_thread / 64 * 64;  // <= 64 MB of scratch
    threads = threads < 64 ? 64 : threads > 4096 ? 4096 : threads;
    *n_blocks = (uint32_t)(threads / 64);
    *scratch_bytes = threads * per_thread;
}

// ---- chained forward kernel (phmm_chain_kernels.hip) ------------------------------------------------
constexpr int CHAIN_MAX_READS = 64;  // reads per chain (one lane per read when the stream offsets are scanned)
struct ChainItem {
    uint32_t region;                // region index
    uint16_t quad;                  // haplotype group ((64/L)/streams haplotypes) inside the region
    uint8_t k, streams;             // columns per lane of this item's body; 1 | 2 | 4 sub-runs swept side by side (L = 16)
    uint32_t read_begin, read_end;  // global read indices [begin, end), all of that region
};
static_assert(sizeof(ChainItem) == 16, "work item record");
struct ChainParams {
    ForwardParams f;
    const ChainItem *items;
    uint32_t n_items;
    uint8_t *redo;     // f32-first kernel only: [n_reads] flags, set where the f64 per-read kernel has to redo a read
};
// ---- shared haplotype prefixes (the chained body's PARK / SUFFIX modes: built and measured in round 4 -- x 0.96-1.10 effective,
// NOTEBOOK 18.4 -- and no longer instantiated: the host side, the API call and the kernels went in round 6) ---------------------
// Columns left of the first base where a haplotype differs from its region's first haplotype (the "trunk") hold the same
// M / I / D for every read -- what the reference's scalar arm skips through find_first_position_where_haplotypes_differ
// (pair_hmm.rs:452-464, 706-717).  Here: the trunk's item PARKS, for every stream row, M~ and D' of the last column of
// the lanes in front of which a group of sharers starts (16 bytes per row; I^ follows from them); a sharers' item sweeps
// the SUFFIX alone -- four haplotypes from the same lane boundary of the trunk on, re-blocked to K' = ceil(suffix / 16)
// columns per lane -- and its first lanes take the parked column as their left neighbour.  Every cell is computed by the
// same operations in the same order as in the full sweep, so results are bit-identical.  Items of both kinds name their
// haplotypes (the planner sorts a region's haplotypes by how much they share); one stream of reads per item.
enum : int { CHAIN_PLAIN = 0, CHAIN_PARK = 1, CHAIN_SUFFIX = 2 };
struct ChainItemX {
    ChainItem it;        // streams == 1; `quad` is not used
    uint16_t hap[4];     // the haplotypes of the wave's four slots (index inside the region; 0xffff: none).  PARK: hap[0] is the trunk
    uint32_t park_row0;  // PARK: first row of this item's block of the parking area (16-byte rows: boundary b and ring position Q at
                         // park_row0 + b park_rows + Q); SUFFIX: the same with its boundary's b park_rows already added
    uint16_t park_rows16;   // ring positions per boundary, in units of 16
    uint16_t mask_or_col0;  // PARK: bit l = the last column of lane l of slot 0 is parked (boundary index = how many lower bits are
                            // set); SUFFIX: first haplotype column of the item
};
static_assert(sizeof(ChainItemX) == 32, "work item record");
// The K ranges a mixed launch is cut into: one kernel per range holds only that range's bodies (phmm_chain_kernels.hip).
#define PHMM_CHAIN_RANGES(X) X(0, 2, 9) X(1, 10, 15) X(2, 16, 19) X(3, 20, 25)
constexpr int kChainRanges = 4;
constexpr int chain_range_of(int k) { return k <= 9 ? 0 : k <= 15 ? 1 : k <= 19 ? 2 : 3; }
// all chained classes of one lanes-per-pair value and one K range in ONE launch (every item carries its K and stream
// count); single_k = the K all items share (the per-K kernel is used), or -(range + 1) for a mixed launch of that range
hipError_t launch_chain(int L, int single_k, const ChainParams &p, hipStream_t stream);  // L lanes per pair: 16, 32 or 64
hipError_t launch_chain_f32(int L, int single_k, const ChainParams &p, hipStream_t stream);  // phmm_chain32_kernels.hip, L = 16 | 32
int chain_max_k();  // largest instantiated K

// ---- engine-level steps (phmm_engine_kernels.hip) -------------------------------------------------
struct PrepParams {
    uint32_t n_reads;
    const uint32_t *read_off;
    const uint8_t *read_bases, *base_q, *ins_q, *del_q;  // originals; ins_q / del_q may be null (flat default)
    const uint8_t *mapq;                                 // [n_reads]
    const uint8_t *pcr_cache;                            // [101] or null when the PCR model is None
    uint8_t *out_q, *out_ins, *out_del, *out_gcp;        // modified copies the forward kernel reads
    double *threshold;                                   // [n_reads] read-disqualification threshold
    uint32_t lds_rows;                                   // >= longest read, multiple of 8 (17 B of LDS per row)
    uint32_t waves_per_read;                             // >= 1: wave c of a read takes positions 64 c + lane, then strides on
    uint32_t default_indel_qual, constant_gcp, base_quality_score_threshold, disable_cap_to_mapq,
        dynamic_disqualification;
    double read_disqualification_scale, expected_error_rate_per_base;
    // small calls (phmm_region_compute): the pre-step reads its inputs straight from the caller's pinned mirror while
    // further blocks of the SAME launch copy the staged block to the device for the kernels behind it (one launch and
    // one PCIe round trip less than stage-in kernel + pre-step); stage_n16 == 0: nothing to copy
    const void *stage_src;
    void *stage_dst;
    uint32_t stage_n16;    // 16-byte units
};
struct PostParams {
    uint32_t n_reads;
    const uint32_t *read_region, *region_read_off, *region_hap_off;
    const uint64_t *out_off;
    const int32_t *region_ref_hap;  // [n_regions] reference haplotype index inside the region, -1 = none; may be null
    double *out;                    // [region][read][hap], normalised in place ...
    double *out_final;              // ... or (small calls) read from `out` and stored here, the caller's pinned mirror; null = in place
    const double *threshold;        // [n_reads]
    uint8_t *keep;                  // [n_reads] 1 = evidence survives filter_poorly_modeled_evidence
    const uint32_t *status_in;      // small calls: the forward kernels' status word, final by now, is handed on to
    uint32_t *status_out;           // the mirror as well (both null otherwise)
    double max_likelihood_difference_cap;
    uint32_t symmetric;
};
hipError_t launch_prep(const PrepParams &p, hipStream_t stream);
hipError_t launch_post(const PostParams &p, hipStream_t stream);
// best allele per read, ties by priority (AlleleLikelihoods::search_best_allele, allele_likelihoods.rs:457-554)
constexpr uint32_t SW_NO_REFERENCE = 0xffffffffu;  // ref_index value: this alignment is skipped (evidence removed / no allele)
struct BestParams {
    uint32_t r_begin, n_reads, n_regions;  // this launch: reads [r_begin, n_reads)
    const uint32_t *region_read_off, *region_hap_off;
    const uint64_t *out_off;
    const double *likelihoods;     // per region row-major [read][hap], as phmm_engine_compute returns them
    const uint8_t *keep;           // [n_reads] or null: 0 = the evidence was removed, no best allele
    const int32_t *priority;       // [n_haps] or null (no tie breaking)
    double threshold;              // get_informative_threshold (:309-315)
    int32_t *best_allele;          // [n_reads] index inside the region, -1 = none
    double *likelihood, *confidence;
    uint32_t *ref_index;           // [n_reads] or null: region_hap_off[g] + best, SW_NO_REFERENCE where there is none
};
hipError_t launch_best_alleles(const BestParams &p, hipStream_t stream);
// post-step and best alleles of the same reads in ONE launch (phmm_region_compute): thread r normalises its row, decides
// keep[r] and goes straight on to the best-allele search over the row it has just written -- the likelihood matrix and the
// keep flags never leave the device between compute_read_likelihoods and realign_reads_to_their_best_haplotype
// (haplotype_caller_engine.rs:1311-1357).  post.n_reads == best.n_reads, best.r_begin == 0, best.likelihoods == post.out,
// best.keep == post.keep.
struct PostBestParams {
    PostParams post;
    BestParams best;
    uint32_t skip_single_allele;   // 1: a region with exactly one haplotype is not realigned (the caller returns before
                                   // realign_reads_to_their_best_haplotype, :1339-1345): ref_index = SW_NO_REFERENCE there
    uint8_t *keep_final;           // small calls: keep flags stored into the caller's pinned mirror as well (or null)
};
hipError_t launch_post_best(const PostBestParams &p, hipStream_t stream);

// The instantiated K values (for every L in {16,32,64}); the planner rounds K up to one of these.
extern const int kInstantiatedK[];
extern const int kNumInstantiatedK;

}  // namespace phmm
