// Smith-Waterman (phmm_sw_kernels.hip): kernel parameters and launchers, shared by the kernel file and phmm_sw.cpp.
#pragma once

#include "phmm_internal.hpp"

namespace phmm {

constexpr int PHMM_SW_STRATEGY_SOFTCLIP = 0, PHMM_SW_STRATEGY_INDEL = 1, PHMM_SW_STRATEGY_LEADING_INDEL = 2,
              PHMM_SW_STRATEGY_IGNORE = 3;  // == PHMM_SW_* of include/phmm.h
// the status block: one word per condition, set by plain stores (no atomics: for small calls the block lies in pinned
// host memory), then [2], [3] = shader clocks / 100 MHz ticks of block 0
constexpr int SW_STATUS_EMPTY = 0;     // an empty reference or alternate sequence
constexpr int SW_STATUS_CAPACITY = 1;  // some CIGAR did not fit its slot (n_cigar holds the size it needs)
// Flag dwords a lane stores per step for its K cells (four bits each): a pair -- candidate tags, gap-open bits -- per 16
// cells, and for a remainder of at most 8 cells ONE dword with the tags in its top and the gap bits in its bottom half.
constexpr int sw_flag_words(int K) { return 2 * (K / 16) + (K % 16 == 0 ? 0 : K % 16 <= 8 ? 1 : 2); }
constexpr int sw_tag_words(int K) { return (K + 15) / 16; }  // the tags-only sweep (SW_LITE): two bits per cell

struct SwParams {
    uint32_t a_begin, n_alignments;        // this launch aligns [a_begin, n_alignments)
    const uint32_t *ref_off, *alt_off;     // [references + 1], [n_alignments + 1]
    const uint32_t *ref_index;             // [n_alignments] reference of each alignment (SW_NO_REFERENCE: skipped), or null: alignment a has reference a
    const uint8_t *ref_bases, *alt_bases;
    int32_t w_match, w_mismatch, w_open, w_extend;
    int strategy;
    const uint64_t *cigar_off;             // [n_alignments + 1], or null: alignment a owns the slot [a * cigar_slot, (a + 1) * cigar_slot)
    uint32_t cigar_slot;
    const uint32_t *alt_clip;              // [2 * n_alignments] or null: (leading, trailing) bases of alternate a that are left out
                                           // (soft clips: the reference aligns the read minus its soft clips, alignment_utils.rs:47-50)
    uint32_t *cigar, *n_cigar;
    int32_t *alignment_offset;
    uint32_t *slab;                        // backtrack flags, one slab per block
    size_t slab_stride;                    // dwords per slab: strips * (max_ref + 16) steps * 2 ceil(K / 16) words * 64 lanes
    uint32_t *status;
    uint32_t max_ref, max_alt;             // longest sequences of the batch
    uint32_t lds_ref_bytes, lds_alt_bytes; // LDS reserved for the two sequences (multiples of 16)
    uint32_t lds_group_bytes;              // LDS of one alignment
    uint32_t groups_per_block;             // alignments a block works on side by side: 64 / L, or 1 for very long sequences
    // two passes (SW_LITE): the tags-only launch lists the alignments whose walk met a gap (todo_out, *todo_out_count: device
    // memory, the counter zeroed beforehand); the full launch behind it aligns todo[0 .. *todo_count) instead of a range
    uint32_t *todo_out, *todo_out_count;
    const uint32_t *todo, *todo_count;
    uint32_t *feedback;                    // or null: where the full launch leaves *todo_count for the host (pinned memory)
    uint32_t todo_min, todo_max;           // a launch over the list runs only if todo_min <= *todo_count <= todo_max (0 = no upper bound):
                                           // a short list goes to the instance with one alignment per wave, a long one to the batch's own
    // every read against every haplotype of its region (phmm_region_compute, small calls: the aligner runs beside the
    // PairHMM kernels and the best allele picks its slot afterwards): pair_stride > 0 = slots per read (>= the largest
    // haplotype count of a region); alignment a is read a / pair_stride against haplotype a % pair_stride of that read's
    // region (no such haplotype: an empty CIGAR); ref_index is not looked at
    uint32_t pair_stride;
    uint32_t pair_single_nh;               // > 0: the call is ONE region with this many haplotypes (read_region / region_hap_off are not read)
    uint32_t *done_counter;                // or null: every block adds one when it has stored its last result (behind a device-scope
                                           // release): the kernel that consumes the alignments waits for the count instead of for an event
    uint32_t report_clock;                 // 1: block 0 stores shader clocks / 100 MHz ticks into status[2..3] (measurement: PHMM_TRACE or the
                                           // switch "sw_clock"; phmm_sw.cpp only, whose status block is the call's own until it returns);
                                           // 2 (tests, switch region_debug_pick bit 1): two words stored LATE, behind the count -- the canary's negative control
    uint32_t pad_was_priority;             // (a wave-priority A/B of round 4 lived here)
    const uint32_t *read_region, *region_hap_off;
    unsigned char *ext;                    // sequences too long for everything to fit LDS: the bottom row and the strip edges of
    size_t ext_stride;                     // block b live at ext + b * ext_stride (device memory), LDS holds the two sequences only
};
// L lanes per alignment (8 / 16 / 32 / 64), K columns per lane (one of kSwK<L>), 64 / L alignments per block
// `wide`: the instance for weights beyond the x4 range (scores as they are, the reference's comparisons and clamp): L = K = 16
enum : int { SW_PLAIN = 0, SW_WIDE = 1, SW_EXT = 2, SW_LITE = 4 };  // kernel variants: un-scaled scores with the reference's clamp; rows in device
                                                                    // memory (the two combine); the tags-only sweep of an ordinary instance
hipError_t launch_sw(int L, int K, bool transposed, int variant, const SwParams &p, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream);
hipError_t launch_sw_gather(const uint32_t *todo, const uint32_t *todo_count, const uint32_t *n_cigar, const int32_t *alignment_offset,
                            const uint32_t *cigar, const uint64_t *cigar_off, uint32_t cap, uint32_t max_entries, uint32_t *out, hipStream_t stream);
int sw_blocks_per_cu(int L, int K, size_t lds_bytes, bool transposed, int variant);  // what a CU holds at once (registers, LDS); 0 on failure
extern const int kSwK16[], kSwK8[], kSwK32[], kSwK64[], kSwK64T[];  // (T: rows per lane of the sweep along the alternate)
extern const int kNumSwK16, kNumSwK8, kNumSwK32, kNumSwK64, kNumSwK64T;

}  // namespace phmm
