// Projection of read -> haplotype alignments onto the reference (phmm_cigar_kernels.hip): kernel parameters and status
// codes, shared by the kernel file and phmm_cigar.cpp.
#pragma once

#include "phmm_internal.hpp"

namespace phmm {

// per-read status (== PHMM_PROJECT_* of include/phmm.h)
constexpr int CIGAR_OK = 0;                 // realigned: new position and cigar are valid
constexpr int CIGAR_UNCHANGED = 1;          // no best allele, or alignment_offset == -1: the read stays as it is
constexpr int CIGAR_ERR_ORDER = -1;         // CigarBuilder: "Cigar has already reached its right (hard) clip"
constexpr int CIGAR_ERR_SOFT_CLIPPED = -2;  // CigarBuilder: "Cigar is completely soft clipped"
constexpr int CIGAR_ERR_NONE = -3;          // CigarBuilder: "Last element cannot be None at this point"
constexpr int CIGAR_ERR_EMPTY = -4;         // CigarBuilder: "No cigar elements left after removing leading and trailing deletions."
constexpr int CIGAR_ERR_PANIC = -5;         // an assert! / panic! of the reference (read past the reference, cigar does not cover the read, ...)
constexpr int CIGAR_ERR_WORKSPACE = -6;     // more elements than the workspace reserves (never with the sizes the host side derives)

struct ProjectParams {
    uint32_t r_begin, n_reads, n_regions;   // this launch projects reads [r_begin, n_reads)
    const uint32_t *region_read_off, *region_hap_off;
    const uint32_t *read_off;          // [n_reads + 1] the reads minus their soft clips
    const uint8_t *read_bases;
    const uint32_t *hap_off;           // [n_haps + 1]
    const uint8_t *hap_bases;
    const int32_t *region_ref_hap;     // [n_regions] reference haplotype inside the region
    const uint64_t *region_reference_start;  // [n_regions]
    const uint32_t *hap_cigar_off, *hap_cigar;   // [n_haps + 1], elements
    const uint32_t *hap_start_wrt_ref;           // [n_haps]
    const int32_t *best_allele;        // [n_reads]
    const uint64_t *sw_cigar_off;      // [n_reads + 1], or null: read r owns the slot [r * sw_cigar_slot, (r + 1) * sw_cigar_slot)
    uint32_t sw_cigar_slot;
    uint32_t sw_pair_stride;           // > 0: the aligner ran every read against every haplotype of its region (SwParams::pair_stride):
                                       // read r's alignment is slot r * sw_pair_stride + best allele (sw_cigar_off must be null)
    const uint32_t *read_clip;         // [2 * n_reads] or null: (leading, trailing) soft-clipped bases inside read_bases that were not aligned
    const uint32_t *ref_index;         // [n_reads] or null: SW_NO_REFERENCE = the read was not aligned (it stays as it is)
    const uint32_t *sw_cigar, *n_sw_cigar;
    const int32_t *sw_offset;
    const uint32_t *orig_cigar_off, *orig_cigar;  // [n_reads + 1], elements
    const uint64_t *out_cigar_off;     // [n_reads + 1]
    uint32_t *out_cigar, *n_out_cigar;
    int64_t *new_pos;
    int32_t *status;
    uint32_t *flags;                   // [0] bit 0: some cigar did not fit its slot; [1] (phmm_pick_reads) != 0: the wait for the aligner ran out of time
    uint32_t *workspace;               // [n_reads - r_begin][4][capacity]
    uint32_t capacity;
    // phmm_pick_reads only: the aligner ran on another stream -- wait until *wait_counter has reached wait_target (its blocks
    // count themselves in behind a release, SwParams::done_counter; compared modulo 2^32) instead of for an event
    const uint32_t *wait_counter;
    uint32_t wait_target;
    uint32_t wait_ticks;               // how long a lane waits at most, in ticks of the 100 MHz clock (wall_clock64); then flags[1] = 1
    // The call's LAST kernel, results straight into the pinned mirror: every block counts itself in behind a release
    // (*finish_counter, device memory, never reset), and the one that brings the count to finish_target stores 1 into
    // *finish_flag -- a word of the mirror the calling thread polls instead of waiting in hipStreamSynchronize (the runtime
    // reports a finished kernel ~6 us after its last store: tools/ubench/sync_latency.hip).  finish_target is set by the launch.
    uint32_t *finish_counter;
    uint32_t finish_target;
    uint32_t *finish_flag;
};
hipError_t launch_project(const ProjectParams &p, hipStream_t stream, uint32_t *blocks = nullptr);
// phmm_post_best_reads and phmm_project_kernel of the same reads as one launch (phmm_pick_reads)
hipError_t launch_pick(const PostBestParams &pb, const ProjectParams &p, hipStream_t stream, uint32_t *blocks = nullptr);
// `bytes` (rounded up to 16; both buffers are 256-aligned and padded) from pinned host memory, by its device address, to `dev`
hipError_t launch_stage_in(const void *host_as_device, void *dev, size_t bytes, hipStream_t stream);

// CigarUtils::calculate_cigar behind the padded alignments (phmm_calculate_cigar): one lane per haplotype
constexpr int CIGAR_SW_FAILURE = 1;  // is_s_w_failure: the reference returns None
constexpr uint32_t SW_PAD_BASES = 10;  // SW_PAD = "NNNNNNNNNN" (cigar_utils.rs:11)
struct CalcParams {
    uint32_t n;
    const uint32_t *ref_off, *alt_off;       // [n + 1] the PADDED sequences (pad + bases + pad)
    const uint8_t *ref_bases, *alt_bases;
    const uint64_t *sw_cigar_off;            // [n + 1] the alignments of the padded sequences, where the aligner left them
    const uint32_t *sw_cigar, *n_sw_cigar;
    const int32_t *sw_offset;
    const uint64_t *out_cigar_off;           // [n + 1]
    uint32_t *out_cigar, *n_out_cigar;
    int32_t *status;                         // 0 a cigar, 1 None (is_s_w_failure), < 0 where the reference panics
    uint32_t *flags;                         // bit 0: some cigar did not fit its slot
    uint32_t *workspace;                     // [n][3][capacity]
    uint32_t capacity;
};
hipError_t launch_calculate_cigar(const CalcParams &p, hipStream_t stream);

}  // namespace phmm
