// Private to the host side of libphmm.so (phmm_api.cpp, phmm_submit.cpp): the engine handle, its staging arenas and the
// two internal entry points the cross-thread queue builds on.  Nothing here is part of the ABI (include/phmm.h).
#pragma once
#include "../../include/phmm.h"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

// Grow-only device arena with a pinned host mirror at identical offsets.  phmm_compute() places the
// whole batch (offset arrays, work lists, payload, status word, results) in it, so a call costs one
// H2D copy, the kernel launches and one D2H copy -- no hipMalloc/hipFree once the arena is warm.
struct Arena {
    char *dev = nullptr, *host = nullptr;
    size_t cap = 0, used = 0;
    double *rescue = nullptr;  // scratch of phmm_rescue for batches staged in this arena (grown on demand)
    size_t rescue_cap = 0;
    // PHMM_MIRROR_CANARY (debug): the result block of the last zero-copy call staged here was poisoned when the call returned
    // and must still be poison when the arena is staged again; and the inputs a zero-copy call staged must still be what was
    // staged when the call ends -- a device store that lands in the pinned mirror after (or outside) its call fails loudly
    size_t canary_off = 0, canary_bytes = 0;
    std::vector<unsigned char> canary_inputs;
};

// Developer switches: read from the PHMM_* environment variables ONCE, in phmm_create, and
// changed afterwards only through phmm_set_switch -- nothing on the host path calls getenv.
struct Switches {
    int force_L = 0;            // PHMM_FORCE_L: 16 / 32 / 64 lanes per pair, 0 = planner's choice
    int force_chain = -1;       // PHMM_FORCE_CHAIN: reads per run of the chained kernel, 0 = per-read kernel only, -1 = planner
    int force_streams = 0;      // PHMM_FORCE_STREAMS: 1 / 2 / 4 streams of the chained kernel
    int no_pipeline = 0;        // PHMM_NO_PIPELINE: host path in one shot whatever the size
    int no_rescue = 0;          // PHMM_NO_RESCUE: leave results below kRescueBelow as the fast kernels made them (A/B only)
    int trace = 0;              // PHMM_TRACE: plan and host-path timing on stderr
    int sw_lite = -1;           // PHMM_SW_LITE: the tags-only first pass of the aligner -- -1 where it pays (adaptive), 0 never, 1 always
    int sw_lanes = 0;           // PHMM_SW_LANES: 8 / 16 / 32 / 64 lanes per Smith-Waterman alignment (0 = by the batch)
    int sw_chunks = 0;          // PHMM_SW_CHUNKS: pieces a phmm_sw_align call is pipelined in (0 = by size, at most 4)
    int sw_transpose = -1;      // PHMM_SW_TRANSPOSE: 0 = small calls never sweep along the alternate sequence, 1 = whenever possible, -1 = by cost
    int sw_clock = 0;           // PHMM_SW_CLOCK: the aligner's block 0 reports the shader clock it ran at (phmm_get_stat "sw_clock_mhz"; also with PHMM_TRACE)
    int sw_no_zero_copy = 0;    // PHMM_SW_NO_ZERO_COPY: small one-piece calls fetch their results by copies like large ones (A/B only)
    int region_flag_wait = 1;   // PHMM_REGION_FLAG_WAIT: 0 = a small region call's thread waits in hipStreamSynchronize instead of polling the
                                // word its last kernel stores into the pinned mirror (A/B)
    int region_pick_timeout_us = 5000;  // PHMM_REGION_PICK_TIMEOUT_US: how long phmm_pick_reads waits for the all-pairs aligner on the other
                                // stream before it gives up (the host then runs the call again the chained way)
    int region_debug_pick = 0;  // PHMM_REGION_DEBUG_PICK (tests): 1 = the all-pairs aligner is enqueued BEHIND phmm_pick_reads on the call's own
                                // stream (both on one hardware queue, in order: the wait can only run out of time); 2 = the all-pairs aligner
                                // stores two words into the call's status block ~300 us AFTER it has counted itself in (round 4's bug, on purpose);
                                // 4 = a call's answer from the region server counts as lost (what a stalled server looks like to its caller)
    int route_shared = 0;       // PHMM_ROUTE_SHARED (opt-in): while MORE than this many of the caller's handles are alive on a device, the one-shot calls
                                // of private handles go through the device's shared combiner (its lanes) instead of their own streams -- 0 = never
    int mirror_canary = 0;      // PHMM_MIRROR_CANARY: 1 = late / stray device stores into the pinned mirror fail the call (Arena::canary_*), 2 = abort()
    int region_own_queue = 1;   // PHMM_REGION_OWN_QUEUE: 0 = one-enqueue calls stay on the handle's ordinary slot-0 stream (A/B)
    int region_server = -1;     // PHMM_REGION_SERVER: region calls go through the device's resident server (phmm_server.cpp) -- -1: the one-shot calls of
                                // PRIVATE handles while more than five of the caller's handles are alive on the device (a handle whose other switches
                                // were changed keeps the launched pipeline); 0 never; 1 every call the server's limits admit, a shared handle's too
    int server_idle_us = 200;   // PHMM_SERVER_IDLE_US: how long the server stays on the chip with nothing in flight and nothing arriving
    int server_trace = 0;       // PHMM_SERVER_TRACE: every task leaves a record (tools/server_trace.py)
    int region_sw_all = -1;     // PHMM_REGION_SW_ALL: a small phmm_region_compute call aligns every read against EVERY haplotype beside the
                                // PairHMM kernels (the best allele picks afterwards) -- -1 up to 2 048 pairs, 0 never, n > 0 up to n pairs
};

constexpr int kSlots = 3;  // pipeline depth of the chunked host path
constexpr int kMaxDevices = 64;  // per-device tables of the process (queue pool, calls in flight); phmm_create refuses ids beyond


struct phmm_handle {
    // Each slot is an independent (arena, stream) pair; single calls use slot 0, the chunked large-batch
    // path rotates through all of them so that staging / H2D of chunk i+1 overlaps the kernels of chunk i.
    Arena arenas[kSlots];
    hipStream_t streams[kSlots] = {nullptr, nullptr, nullptr};
    int slot = 0;
    Arena &A() { return arenas[slot]; }
    hipStream_t S() { return streams[slot]; }
    hipStream_t stream0_ordinary = nullptr;  // streams[0] is this or swork.pair_main[0] (latch_slot0)
    // the chained launches of one batch run side by side: the first on the caller's stream, the others here (phmm_batch_launch)
    static constexpr int kSideStreams = 3;
    hipStream_t side_streams[kSideStreams] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kSideStreams] = {};
    int device = 0;
    unsigned flags = 0;
    bool internal = false;     // a lane of a shared handle, or a device's backing handle (route_shared): not one of the caller's own
    bool sw_touched = false;   // phmm_set_switch was called on this handle: its calls stay on its own resources (A/B runs, tests)
    void *server = nullptr;    // the device's resident region server (phmm_server.cpp), once this handle has asked for it
    phmm_handle *backing = nullptr;  // the shared handle of (device, flags) this handle's small calls go through while many are alive
    double *d_eps = nullptr, *d_eps_mis = nullptr, *d_mm = nullptr, *d_ratio_mis = nullptr, *d_inv_om = nullptr;
    uint8_t *d_pcr_cache = nullptr;  // [4][128]: PCR indel model caches, one row per model
    std::string err;
    int err_code = PHMM_OK;  // status of the last failure (set together with err)
    Switches sw;
    struct SwWork {  // phmm_sw_align (phmm_sw.cpp): grow-only staging and backtrack slabs
        char *dev = nullptr, *host = nullptr;
        char *host_dev = nullptr;  // the pinned mirror as the device sees it (small calls: kernels store their results there)
        size_t cap = 0;
        uint32_t *slab = nullptr;
        size_t slab_bytes = 0;
        uint32_t *ws = nullptr;  // the projection's builders (phmm_realign_reads)
        size_t ws_bytes = 0;
        hipEvent_t ev_second = nullptr;  // the one second pass of a call in pieces is done
        int lite_skip = 0;             // calls that go straight to the full Smith-Waterman instance (the last two-pass call met too many gaps)
        uint64_t last_second_pass = 0; // alignments of the last call that the full instance had to align again (phmm_get_stat "sw_second_pass")
        unsigned char *ext = nullptr;  // bottom rows / strip edges of alignments too long for LDS (SwGeometry::ext_stride)
        size_t ext_bytes = 0;
        static constexpr int kMaxChunks = 8;      // pieces of one call: piece c+1 is staged and copied while piece c computes
        hipEvent_t ev_in[kMaxChunks] = {}, ev_out[kMaxChunks] = {}, ev_k0[kMaxChunks] = {}, ev_k1[kMaxChunks] = {};  // inputs landed; results landed; around each kernel
                                                   // (phmm_get_stat "sw_kernel_us" = the kernels' own time, summed)
        uint64_t last_kernel_us = 0, last_backtrack_bytes = 0, last_clock_mhz = 0;
        hipStream_t all_stream[2] = {};       // phmm_region_compute, small calls: the aligner over every (read, haplotype) pair runs
        uint32_t *d_pair_done = nullptr;      // here, beside the PairHMM kernels; its blocks count themselves in here when done, and
        uint32_t pair_done_target = 0;        // phmm_pick_reads on the other stream waits for the count (never reset: compared modulo 2^32)
        uint32_t finish_count = 0;            // blocks of last kernels that have counted / will count themselves in at d_pair_done[16]
                                              // (ProjectParams::finish_counter)
        hipStream_t pair_main[2] = {};        // ... and the other kernels of such a call here (hardware queues of their own; [1]: the
                                              // pair whose two streams own disjoint halves of the CUs, phmm_region.cpp)
        int queue_index = -1;                 // >= 0: pair_main[0] / all_stream[0] are this handle's pair of the device's queue pool (queues_acquire)
        uint64_t region_sw_all_calls = 0;     // how many calls went that way (phmm_get_stat "region_sw_all")
        uint64_t region_pick_timeouts = 0;    // ... and how many of them were run again as the chain because phmm_pick_reads' wait ran
                                              // out of time (phmm_get_stat "region_pick_timeouts")
        hipEvent_t region_sw_done = nullptr;  // phmm_region_compute in chunks: the slab and the workspace are one per handle, so
        bool region_sw_pending = false;       // a chunk's alignment kernels wait for those of the chunk before it
        std::unordered_map<uint64_t, int> blocks_per_cu;  // by (lanes, columns, LDS bytes): asked of the runtime once
    } swork;
    uint64_t stat_staged_bytes = 0;   // payload bytes copied into pinned staging by this handle (phmm_get_stat)
    uint64_t stat_rescue_passes = 0;  // how many batches needed the exact pass (phmm_get_stat)
    struct Combiner *comb = nullptr;  // phmm_submit / phmm_wait state, created by the first phmm_submit
    uint32_t gpu_sharers = 1;         // flows computing at the same time (phmm_wait, combined flushes): the planner stops
                                      // trading lanes for waves once the batch fills its share of the chip
    uint32_t busy_lanes = 1;          // a lane of a shared handle: lanes computing right now, this one included (phmm_wait)
    bool defer_d2h = false;           // see eager_d2h(): set around pipelined chunks and combined flushes
    std::once_flag comb_once;
};

namespace phmm_host {

// Inputs of several independent submissions that one launch computes together (phmm_submit): part s contributes
// read_bytes[s] bytes to each of the five per-base read arrays, hap_bytes[s] to the haplotype bases, and receives
// n_out[s] results.
struct Parts {
    std::vector<const uint8_t *> src[6];
    std::vector<size_t> read_bytes, hap_bytes;
    std::vector<double *> out;
    std::vector<uint64_t> n_out;
    std::vector<uint32_t> first_region;  // [parts + 1] regions of the combined batch each part owns
};

struct PendingCompute {
    phmm_batch *b = nullptr;
    int slot = 0;
    hipStream_t stream = nullptr;  // the stream the batch was enqueued on (slot 0's may change between calls: latch_slot0)
    double *out = nullptr;
    const Parts *parts = nullptr;  // non-null: results go to parts->out[s] instead of `out`
    bool d2h_pending = false;      // the D2H copy of [status | out] is still to be issued (see eager_d2h)
    bool zero_copy = false;        // the kernels wrote `out` into the pinned mirror themselves; status is checked here
};

constexpr size_t kCombineBytes = 4u << 20;  // per-array bytes one combined flush of phmm_wait takes

// Stage one batch in the current slot's arena and enqueue H2D and kernels (and, for a one-shot call, the D2H) on its
// stream; no sync.  finish_compute waits, fetches the results if that is still to do, and hands them to the caller.
int enqueue_compute(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                    const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q,
                    const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off, const uint8_t *hap_bases,
                    const uint64_t *out_off, double *out, PendingCompute *pending, const Parts *parts = nullptr);
int finish_compute(phmm_handle *h, PendingCompute *p);

// phmm_compute restricted to the regions [g_begin, g_end) / to the listed regions of the caller's validated arrays
// (phmm_compute_multi: one range or one list per engine, nothing is gathered first).
int compute_range(phmm_handle *h, uint32_t g_begin, uint32_t g_end, const uint32_t *region_read_off,
                  const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                  const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off,
                  const uint8_t *hap_bases, const uint64_t *out_off, double *out);
int compute_list(phmm_handle *h, const uint32_t *list, uint32_t n_list, const uint32_t *region_read_off,
                 const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                 const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off,
                 const uint8_t *hap_bases, const uint64_t *out_off, double *out);

// ---- internals the per-region pipeline (phmm_region.cpp) builds on ------------------------------------------------------
// Plan a batch (shape classes, work lists) with its device metadata placed in the current slot's arena; `extra_arena_bytes`
// of room are reserved behind it for the caller's own staging (phmm_api.cpp).
phmm_batch *batch_create_in_arena(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                                  const uint32_t *read_off, const uint32_t *hap_off, const uint64_t *out_off, size_t extra_arena_bytes);
struct BatchView {  // what a caller that launches its own kernels around the forward kernels needs to know of a batch
    uint32_t n_regions, n_reads, n_haps, max_h;
    uint64_t n_out, read_bytes, hap_bytes;
    bool tight_out;
    const uint32_t *d_read_region, *d_region_read_off, *d_region_hap_off, *d_read_off, *d_hap_off;
    const uint64_t *d_out_off;
};
BatchView batch_view(const phmm_batch *b);
void batch_set_status(phmm_batch *b, uint32_t *d_status);                       // the device status word the forward kernels raise
bool batch_set_inline_rescue(phmm_handle *h, phmm_batch *b);                    // the exact pass rides behind the forward kernels (arena scratch)
void batch_copy_out(const phmm_batch *b, const double *src, double *out);        // results -> caller, gaps of out_off untouched
bool eager_d2h(const phmm_handle *h);
bool zero_copy_allowed(const phmm_handle *h);  // the mirror path (no copies at all): also inside a combined flush

// Regions [g0, g1) of a caller's batch with every offset array rebased to zero: what one pipelined chunk is made of.
struct ChunkView {
    uint32_t g0 = 0, g1 = 0, r0 = 0, r1 = 0, h0 = 0, h1 = 0;
    uint32_t index = 0;  // how many chunks came before this one
    bool started = false;
    bool mixed = false;  // regions of different shapes: a chunk also has to be large enough for the chained kernel
    bool f32_first = false;  // the handle's precision mode (decides the chunk sizes)
    size_t read_byte0 = 0, hap_byte0 = 0;
    std::vector<uint32_t> rro, rho, ro, ho;
    std::vector<uint64_t> oo;
};
bool next_chunk(ChunkView &c, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                const uint32_t *read_off, const uint32_t *hap_off, const uint64_t *out_off, bool whole = false);
size_t one_shot_bytes();     // per-array bytes up to which a host-buffer call goes in one shot
// Hardware queues of a handle's own for its small region calls (phmm_region.cpp): swork.pair_main[0] / all_stream[0].
bool queues_acquire(phmm_handle *h);
void queues_release(phmm_handle *h);  // (phmm_destroy)
bool halves_acquire(phmm_handle *h);  // swork.pair_main[1] / all_stream[1]
void halves_release(phmm_handle *h);
void handle_born(phmm_handle *h);     // (phmm_create / phmm_destroy: handles alive on the device)
void handle_died(phmm_handle *h);
// At the top of a call: slot 0's stream is the handle's own queue while at most four handles live on the device (beyond that
// the queues would share the command processor's four pipes pairwise, and the runtime's own multiplexing of ordinary streams
// does better: 8 private handles, two calls per region 20.5 k regions/s against 17.5 k), else its ordinary stream.
void latch_slot0(phmm_handle *h);
// Many private handles on one device (a handle per rayon worker, INTEGRATION.md section 4, at --threads 16 / 32): every caller
// thread then sits in its own hipStreamSynchronize and every call is a lone latency-bound chain -- 32 private handles ran at
// HALF the rate of 16 (threads_bench: own 32 threads 24 k regions/s against 47 k; fused 12.5 against 22 k).  Past four live
// handles a private handle's one-shot call is therefore handed to the device's shared handle of the same flags (phmm_submit /
// phmm_wait inside the library: concurrent callers merge into one flush, waiters sleep).  Returns that handle, or null: few
// handles, a lane / backing handle itself, a handle whose switches were changed, or PHMM_ROUTE_SHARED=0.
phmm_handle *route_shared(phmm_handle *h);
phmm_handle *create_internal(int device, unsigned flags);  // phmm_create for lanes and backing handles (not counted as the caller's)
// PHMM_MIRROR_CANARY (no-ops unless the switch is set).  before_staging: the poison of the arena's last zero-copy call is
// intact (false: h->err / err_code are set).  staged: a zero-copy call has staged `in_bytes` of inputs -- keep a copy.
// after_call: the inputs are still what was staged, then poison [res_off, res_off + res_bytes) (false as above).
bool canary_before_staging(phmm_handle *h, Arena &A);
void canary_staged(phmm_handle *h, Arena &A, size_t in_bytes);
bool canary_after_call(phmm_handle *h, Arena &A, size_t res_off, size_t res_bytes);
size_t stage_in_bytes();     // inputs up to this size are fetched from the pinned mirror by a kernel
size_t zero_copy_out_bytes();  // results up to this size are stored into the pinned mirror by the kernels

// Worker geometry of one batch of Smith-Waterman alignments (phmm_sw.cpp).
struct SwGeometry {
    int L = 0, K = 0, per_cu = 0;
    bool transposed = false;
    bool wide = false;  // weights beyond the scaled kernels' range: the un-scaled instance with the reference's clamp
    int variant = 0;    // phmm::SW_WIDE | phmm::SW_EXT: which special instance, 0 = the ordinary ones
    size_t strips = 0, lds_ref = 0, lds_alt = 0, lds_group = 0, gpb = 0, lds = 0, flag_words = 0, slab_stride = 0, max_workers = 0;
    size_t ext_stride = 0;  // > 0: the bottom row and strip edges of a block live in device memory (sequences beyond ~8 000 bases)
};
// (`params`: the weights decide between the scaled kernels and the wide instance, or refuse what overflows 32 bits in the
// reference as well)
int sw_plan(phmm_handle *h, const std::string &who, uint32_t n_alignments, uint32_t max_ref, uint32_t max_alt,
            const phmm_sw_parameters *params, SwGeometry *G);

// The caller's arrays of one phmm_region_compute call (include/phmm.h), or of one chunk / one combined flush of it.
struct RegionArgs {
    phmm_engine_config cfg{};
    phmm_realign_config rcfg{};
    uint32_t n_regions = 0;
    const uint32_t *region_read_off = nullptr, *region_hap_off = nullptr, *read_off = nullptr;
    const uint8_t *read_bases = nullptr, *base_q = nullptr, *ins_q = nullptr, *del_q = nullptr, *mapq = nullptr;
    const uint32_t *read_soft_clip = nullptr;
    const uint32_t *hap_off = nullptr;
    const uint8_t *hap_bases = nullptr;
    const int32_t *region_ref_hap = nullptr;
    const uint64_t *out_off = nullptr;
    const int32_t *hap_priority = nullptr;
    const uint64_t *region_reference_start = nullptr;
    const uint32_t *hap_cigar_off = nullptr, *hap_cigar = nullptr, *hap_start_wrt_ref = nullptr, *orig_cigar_off = nullptr, *orig_cigar = nullptr;
    const uint64_t *out_cigar_off = nullptr;
    double *out = nullptr;
    uint8_t *keep = nullptr;
    int32_t *best_allele = nullptr;
    double *likelihood = nullptr, *confidence = nullptr;
    uint32_t *out_cigar = nullptr, *n_out_cigar = nullptr;
    int64_t *new_pos = nullptr;
    int32_t *status = nullptr;
};
// The resident region server (phmm_server.cpp): submit stages the call in a slot and hands it to the device's server, wait
// polls for its finish word and hands the results over.
struct ServerPending;
constexpr int kServerNotTaken = -2000;  // server_region_submit: the call is outside the server's limits -- nothing was done
constexpr int kServerRedo = -2001;      // server_region_wait: run the call again the launched way (an alignment outgrew its slot)
int server_region_submit(phmm_handle *h, const RegionArgs &a, ServerPending **out, bool via_submit);
int user_handles_on(int device);  // the caller's handles alive on the device (phmm_api.cpp)
// the wait of a one-shot call: hipStreamSynchronize, or -- more caller handles than cores -- looks at the stream between 20 us sleeps
int process_cores();  // what the process may run on: its affinity mask, its container's CPU quota
bool more_callers_than_cores(const phmm_handle *h);
hipError_t wait_stream(const phmm_handle *h, hipStream_t s);
int server_region_wait(phmm_handle *h, ServerPending *p, std::string *err, RegionArgs *redo_args);
uint64_t server_stat(int device, const char *name);
void server_yield(int device);
int region_calls_in_flight(int device);  // phmm_region_compute calls of the launched kind between enqueue and finish (phmm_region.cpp)
void server_quiesce(int device);
// argument check of phmm_region_compute / phmm_region_submit: the message of the first violation, or empty
std::string region_validate(const RegionArgs &a);
// the call itself on validated arguments (phmm_region.cpp); one thread per handle
int region_compute(phmm_handle *h, const RegionArgs &a);
int region_compute_parts(phmm_handle *h, const RegionArgs &combined, const std::vector<RegionArgs> &parts);
int region_compute_range(phmm_handle *h, const RegionArgs &a, uint32_t g0, uint32_t g1);  // regions [g0, g1) of validated arguments
RegionArgs region_pack_args(const phmm_engine_config *cfg, const phmm_realign_config *rcfg, uint32_t n_regions, const uint32_t *region_read_off,
                            const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                            const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *mapq, const uint32_t *read_soft_clip, const uint32_t *hap_off,
                            const uint8_t *hap_bases, const int32_t *region_ref_hap, const uint64_t *out_off, const int32_t *hap_priority,
                            const uint64_t *region_reference_start, const uint32_t *hap_cigar_off, const uint32_t *hap_cigar,
                            const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar, const uint64_t *out_cigar_off,
                            double *out, uint8_t *keep, int32_t *best_allele, double *likelihood, double *confidence, uint32_t *out_cigar,
                            uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status);

// What every entry point checks before it touches the arrays; returns the message of the first violation or nullptr.
const char *validate_offsets(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                             const uint32_t *read_off, const uint32_t *hap_off, const uint64_t *out_off, bool *tight_out);

// phmm_submit / phmm_wait are the only entry points several threads may call on one handle, so their messages are kept
// per calling thread (phmm_last_error returns them); every other entry point resets the pair.
void set_thread_error(const phmm_handle *h, const std::string &msg);
void clear_thread_error(const phmm_handle *h);

void combiner_destroy(Combiner *c);  // phmm_submit.cpp; called by phmm_destroy
void combiner_set_switches(Combiner *c, const Switches &sw);  // phmm_set_switch on a shared handle reaches its lanes
uint64_t combiner_stat(Combiner *c, const char *name);        // sum of phmm_get_stat over the lanes

}  // namespace phmm_host
