// phmm_region_compute (include/phmm.h): one enqueue per region -- or per batch of regions -- for everything the reference
// does with numbers between PairHMMLikelihoodCalculationEngine::compute_read_likelihoods
// (src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs:195-242) and
// AssemblyBasedCallerUtils::realign_reads_to_their_best_haplotype (src/assembly/assembly_based_caller_utils.rs:208-246),
// which follow each other directly in HaplotypeCallerEngine::call_region (src/haplotype/haplotype_caller_engine.rs:1311-1357):
//
//   stage-in | phmm_prep_reads | phmm_forward* (| phmm_rescue) | phmm_post_best_reads | phmm_sw_align_kernel | phmm_project_kernel
//
// all on one stream; the likelihood matrix, the keep flags, the reads' best haplotypes and the read -> haplotype
// alignments stay in device memory from one kernel to the next.  Host side only: validation, staging, launch geometry,
// the chunk pipeline of large calls, status.  No CPU path.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "phmm_cigar_internal.hpp"
#include "phmm_host.hpp"
#include "phmm_region_internal.hpp"
#include "phmm_sw_internal.hpp"

using namespace phmm;
using namespace phmm_host;

namespace {

struct DevGuard {
    int prev = -1, dev;
    explicit DevGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DevGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

bool ok(phmm_handle *h, hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    h->err = std::string(what) + ": " + hipGetErrorString(e);
    h->err_code = PHMM_ERR_HIP;
    return false;
}

struct PendingRegion {
    phmm_batch *b = nullptr;
    int slot = 0;
    hipStream_t stream = nullptr; // the stream the batch was enqueued on
    RegionArgs a;                 // the (chunk's) arguments: where the results go
    const std::vector<RegionArgs> *parts = nullptr;  // non-null: `a` holds the combined offset arrays of several submissions
                                                     // (phmm_region_submit), payload and results are theirs
    Layout L;
    bool d2h_pending = false, zero_copy = false;
    bool lite = false;            // the aligner ran in two passes: the results' header holds how many alignments went round again
    uint32_t sw_capacity = 0;
    bool inline_rescue = false;   // the exact pass below -600 rode behind the forward kernels
    uint32_t pair_stride = 0;     // > 0: the aligner took every read against every haplotype of its region, beside the PairHMM kernels
    const uint32_t *finish_flag = nullptr;  // a word of the pinned mirror the call's last kernel sets when every block of it is through
    hipStream_t all_stream = nullptr;       // pair_stride > 0: the stream the all-pairs aligner was enqueued on
    PendingRegion() = default;
};

constexpr size_t kHalvesUpToPairs = 512;  // (read, haplotype) pairs up to which a call's two streams get half the CUs each
std::atomic<int> g_region_calls[kMaxDevices];  // phmm_region_compute calls between enqueue and finish, per device (all handles of the process)
struct InFlight {
    std::atomic<int> &n;
    explicit InFlight(int device) : n(g_region_calls[device % kMaxDevices]) {
        n.fetch_add(1, std::memory_order_seq_cst);
        phmm_host::server_yield(device);  // (the resident server, if it is on the chip, makes room: its waves fill every SIMD)
    }
    ~InFlight() { n.fetch_sub(1, std::memory_order_relaxed); }
};

// region_finish -> region_one_shot: phmm_pick_reads gave up waiting for the all-pairs aligner (ProjectParams::wait_ticks); both
// streams are idle again, nothing was handed over, the call goes round once more the chained way.  Never leaves this file.
constexpr int kPickTimedOut = -1000;
constexpr uint32_t kFirstSwCapacity = 24;  // CIGAR elements reserved per read -> haplotype alignment (grown and redone when one needs more)

}  // namespace

namespace phmm_host {

int region_calls_in_flight(int device) { return g_region_calls[device % kMaxDevices].load(std::memory_order_seq_cst); }

std::string region_validate(const RegionArgs &a) {
    const std::string w = "phmm_region_compute: ";
    if (a.cfg.pcr_error_model > 3) return w + "Unknown PCR Error Model";  // engine.rs:89
    if (a.rcfg.overhang_strategy < PHMM_SW_SOFTCLIP || a.rcfg.overhang_strategy > PHMM_SW_IGNORE) return w + "unknown overhang strategy";
    if (!(a.rcfg.informative_threshold >= 0.0)) return w + "the informative threshold must be a non-negative number";
    if (const char *bad = validate_offsets(a.n_regions, a.region_read_off, a.region_hap_off, a.read_off, a.hap_off, a.out_off, nullptr))
        return bad;
    const uint32_t ng = a.n_regions, nr = a.region_read_off[ng], nh = a.region_hap_off[ng];
    if ((a.read_off[nr] && (!a.read_bases || !a.base_q)) || (nr && (!a.mapq || !a.keep)) || (a.hap_off[nh] && !a.hap_bases) ||
        (a.out_off[ng] && !a.out))
        return w + "null pointer";
    if (!nr) return "";
    if (!a.region_ref_hap || !a.region_reference_start || !a.hap_cigar_off || !a.hap_start_wrt_ref || !a.orig_cigar_off || !a.out_cigar_off ||
        !a.best_allele || !a.likelihood || !a.confidence || !a.n_out_cigar || !a.new_pos || !a.status)
        return w + "null array";
    if (a.hap_cigar_off[0] != 0 || a.orig_cigar_off[0] != 0 || a.out_cigar_off[0] != 0) return w + "offset arrays must start at 0";
    for (uint32_t g = 0; g < ng; ++g) {
        const uint32_t nrg = a.region_read_off[g + 1] - a.region_read_off[g], nhg = a.region_hap_off[g + 1] - a.region_hap_off[g];
        if (nrg && nhg && (a.region_ref_hap[g] < 0 || (uint32_t)a.region_ref_hap[g] >= nhg))
            return w + "every region with reads and haplotypes needs its reference haplotype (region_ref_hap inside the region)";
    }
    for (uint32_t x = 0; x < nh; ++x)
        if (a.hap_cigar_off[x + 1] < a.hap_cigar_off[x]) return w + "offsets not monotonic";
    for (uint32_t r = 0; r < nr; ++r) {
        if (a.orig_cigar_off[r + 1] < a.orig_cigar_off[r] || a.out_cigar_off[r + 1] < a.out_cigar_off[r]) return w + "offsets not monotonic";
        if (a.read_soft_clip && (uint64_t)a.read_soft_clip[2 * r] + a.read_soft_clip[2 * r + 1] > a.read_off[r + 1] - a.read_off[r])
            return w + "a read's soft clips are longer than the read";
    }
    if ((a.hap_cigar_off[nh] && !a.hap_cigar) || (a.orig_cigar_off[nr] && !a.orig_cigar) || (a.out_cigar_off[nr] && !a.out_cigar))
        return w + "null array";
    return "";
}

}  // namespace phmm_host

namespace {

int fail(phmm_handle *h, const std::string &msg, int code = PHMM_ERR_INVALID_ARG) {
    h->err = msg;
    return h->err_code = code;
}

// grow-only device buffers of the aligner that a handle's slots share (callers see to it that nothing is in flight on them)
bool ensure_sw_buffers(phmm_handle *h, size_t slab_bytes, size_t ws_bytes) {
    phmm_handle::SwWork &W = h->swork;
    if (W.slab_bytes >= slab_bytes && W.ws_bytes >= ws_bytes) return true;
    for (int i = 0; i < kSlots; ++i) (void)hipStreamSynchronize(h->streams[i]);
    W.region_sw_pending = false;
    if (W.slab_bytes < slab_bytes) {
        if (W.slab) (void)hipFree(W.slab);
        W.slab = nullptr;
        W.slab_bytes = 0;
        if (!ok(h, hipMalloc((void **)&W.slab, slab_bytes), "hipMalloc(sw backtrack)")) return false;
        W.slab_bytes = slab_bytes;
    }
    if (W.ws_bytes < ws_bytes) {
        if (W.ws) (void)hipFree(W.ws);
        W.ws = nullptr;
        W.ws_bytes = 0;
        if (!ok(h, hipMalloc((void **)&W.ws, ws_bytes), "hipMalloc(project workspace)")) return false;
        W.ws_bytes = ws_bytes;
    }
    return true;
}

}  // namespace

// ---- hardware queues of a handle's own --------------------------------------------------------------------------------------
// The runtime maps ordinary streams onto four hardware queues by its own rule: four private handles' slot streams were seen
// on THREE of them (rocprofv3 queue ids, `kernels in flight: 3: 76 %, 4: 0 %`), three handles' on two -- their chains of
// dependent kernels then run one behind the other.  A stream created with a CU mask always gets a hardware queue of its own,
// and queues whose ids are equal modulo four share a pipe of the command processor (NOTEBOOK 18.1).  So the device keeps a pool:
// eight full-mask streams created back to back, [o0 o1 o2 o3 a0 a1 a2 a3]; the k-th handle to ask gets o(k mod 4) for its
// calls' kernels and a((k + 2) mod 4) for the all-pairs aligner -- four handles' chains on four pipes, a handle's two queues
// two pipes apart.  A handle takes its pair at its first one-enqueue call (latch_slot0: o becomes its slot-0 stream).  One region per call, chain: 3 threads 13.3 -> 17.4 k regions/s, 4 threads 17.7 -> 21.4 k.
namespace {
constexpr int kHalvesPairs = 4;
constexpr int kPooledHandles = 4;  // ONE batch: a second one (sixteen queues beside the runtime's four and the lanes' halves) is more than the
                                   // command processor keeps mapped -- 30 x 3 regions from 8 threads on the shared handle 42 -> 19 k regions/s
struct QueuePool {
    std::mutex mu;
    std::vector<std::array<hipStream_t, 8>> batches;
    std::vector<int> free_index;
    int next = 0;
    int halves_pairs = 0;
} g_queue_pool[kMaxDevices];
}  // namespace

namespace phmm_host {

bool queues_acquire(phmm_handle *h) {
    phmm_handle::SwWork &W = h->swork;
    if (W.queue_index >= 0) return true;
    if (W.queue_index == -2) return false;  // (refused before)
    QueuePool &P = g_queue_pool[h->device % kMaxDevices];
    std::lock_guard<std::mutex> lk(P.mu);
    int k;
    if (!P.free_index.empty()) {
        k = P.free_index.back();
        P.free_index.pop_back();
    } else {
        k = P.next;
        if (k >= kPooledHandles) {  // (further handles stay on ordinary streams: the runtime's four queues)
            W.queue_index = -2;
            return false;
        }
        if ((size_t)(k / 4) >= P.batches.size()) {
            DevGuard on_device(h->device);  // (hipExtStreamCreateWithCUMask creates on the calling thread's current device)
            int cus = 0;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
            const uint32_t words = (uint32_t)std::min(32, std::max(1, (cus + 31) / 32));
            uint32_t mask[32];
            for (uint32_t w = 0; w < 32; ++w) mask[w] = w + 1 < words || cus % 32 == 0 ? 0xffffffffu : (1u << (cus % 32)) - 1u;
            std::array<hipStream_t, 8> b{};
            for (hipStream_t &s : b)
                if (hipExtStreamCreateWithCUMask(&s, words, mask) != hipSuccess) {
                    for (hipStream_t t : b)
                        if (t) (void)hipStreamDestroy(t);
                    (void)hipGetLastError();
                    W.queue_index = -2;
                    return false;
                }
            P.batches.push_back(b);
        }
        P.next = k + 1;
    }
    W.queue_index = k;
    W.pair_main[0] = P.batches[k / 4][k % 4];
    W.all_stream[0] = P.batches[k / 4][4 + (k + 2) % 4];
    return true;
}

void queues_release(phmm_handle *h) {
    phmm_handle::SwWork &W = h->swork;
    if (W.queue_index < 0) return;
    (void)hipStreamSynchronize(W.pair_main[0]);
    (void)hipStreamSynchronize(W.all_stream[0]);
    QueuePool &P = g_queue_pool[h->device % kMaxDevices];
    std::lock_guard<std::mutex> lk(P.mu);
    P.free_index.push_back(W.queue_index);
    W.queue_index = -1;
    W.pair_main[0] = W.all_stream[0] = nullptr;  // (the streams stay with the pool)
}

// The HALVES pair of a handle (swork.pair_main[1] / all_stream[1]: the call's kernels on one half of the CUs, the all-pairs
// aligner on the other), created back to back; at most four such pairs on a device (more mapped queues than the command
// processor holds at a time cost every call dearly: see kPooledHandles).
bool halves_acquire(phmm_handle *h) {
    phmm_handle::SwWork &W = h->swork;
    if (W.all_stream[1]) return true;
    QueuePool &P = g_queue_pool[h->device % kMaxDevices];
    std::lock_guard<std::mutex> lk(P.mu);
    if (P.halves_pairs >= kHalvesPairs) return false;
    DevGuard on_device(h->device);
    uint32_t mask_a[32], mask_b[32];
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    const uint32_t words = (uint32_t)std::min(32, std::max(1, (cus + 31) / 32));
    for (uint32_t w = 0; w < 32; ++w) {
        const uint32_t all = w + 1 < words || cus % 32 == 0 ? 0xffffffffu : (1u << (cus % 32)) - 1u;
        mask_a[w] = w >= words / 2 && words >= 2 ? 0u : all;
        mask_b[w] = w < words / 2 ? 0u : all;
    }
    hipStream_t m = nullptr, t = nullptr;
    if (hipExtStreamCreateWithCUMask(&m, words, mask_a) != hipSuccess || hipExtStreamCreateWithCUMask(&t, words, mask_b) != hipSuccess) {
        if (m) (void)hipStreamDestroy(m);
        (void)hipGetLastError();
        return false;
    }
    W.pair_main[1] = m;
    W.all_stream[1] = t;
    P.halves_pairs += 1;
    return true;
}

void halves_release(phmm_handle *h) {  // (phmm_destroy, before it destroys the two streams)
    if (!h->swork.all_stream[1]) return;
    QueuePool &P = g_queue_pool[h->device % kMaxDevices];
    std::lock_guard<std::mutex> lk(P.mu);
    P.halves_pairs -= 1;
}

std::atomic<int> g_live_handles[kMaxDevices];
void handle_born(phmm_handle *h) { g_live_handles[h->device % kMaxDevices].fetch_add(1, std::memory_order_relaxed); }
void handle_died(phmm_handle *h) { g_live_handles[h->device % kMaxDevices].fetch_sub(1, std::memory_order_relaxed); }
void latch_slot0(phmm_handle *h) {
    // (taken at the first call, not at phmm_create: a shared handle never makes one itself, its lanes do)
    const bool few = g_live_handles[h->device % kMaxDevices].load(std::memory_order_relaxed) <= 4;
    bool own = false;
    if (few && h->sw.region_own_queue) {
        // (the entry points call this before their own device guard, and hipExtStreamCreateWithCUMask creates on the CALLING
        // THREAD's current device: a worker thread of phmm_*_compute_multi, or a rayon thread of hip_backend.rs, is on device 0
        // whatever the handle's -- the pool's streams must belong to the device they are filed under)
        if (h->swork.queue_index == -1) {
            DevGuard dg(h->device);
            own = queues_acquire(h);
        } else {
            own = h->swork.queue_index >= 0;
        }
    }
    h->streams[0] = own ? h->swork.pair_main[0] : h->stream0_ordinary;
}

}  // namespace phmm_host

namespace {

// Stage one batch of regions in the current slot's arena and enqueue everything on its stream.  No sync.
int region_enqueue(phmm_handle *h, const RegionArgs &a, const std::vector<RegionArgs> *parts, uint32_t sw_capacity, bool chained,
                   bool may_align_all, PendingRegion *pending) {
    const std::string who = "phmm_region_compute";
    const uint32_t ng = a.n_regions, nr = a.region_read_off[ng], nh = a.region_hap_off[ng];
    // A small call leaves most of the chip idle and is a chain of dependent kernels, the aligner the longest of them -- and
    // the aligner needs nothing the PairHMM kernels make except WHICH haplotype a read is aligned to.  So it aligns every
    // read against every haplotype of its region on a stream of its own, beside pre-step and PairHMM, and the kernel behind
    // both takes the best allele's slot: the same alignments as the chain's, Nh times the aligner's work on SIMDs that had
    // none, one kernel less on the critical path (haplotype_caller_engine.rs:1311-1357 is the sequence this replaces).
    uint32_t pair_stride = 0;
    if (may_align_all && !chained && nr && nh && h->sw.region_sw_all != 0) {
        uint32_t max_nh = 0;
        for (uint32_t g = 0; g < ng; ++g) max_nh = std::max(max_nh, a.region_hap_off[g + 1] - a.region_hap_off[g]);
        // (by itself only while this is the one region call in flight in the process: with several callers the chip is not idle,
        // and Nh times the aligner's work comes out of the other calls' time -- 4 threads: 21 k regions/s the plain way, 14 k this way)
        const int in_flight = std::max<int>((int)h->busy_lanes, g_region_calls[h->device % kMaxDevices].load(std::memory_order_relaxed));
        // (... two calls of a few hundred pairs each are still small against the chip: 30 x 3 regions from two threads 141 -> 100 us
        // per call.  From four callers on every call's two queues of its own are more hardware queues than run at a time:
        // 157 -> 218 us.)
        // (... three callers, each on queues of its own: 60 x 4 regions 17.9 -> 21.5 k regions/s, 100 x 5 the same either way)
        const uint64_t limit = h->sw.region_sw_all > 0 ? (uint64_t)h->sw.region_sw_all
                               : in_flight <= 1 ? 2048u : in_flight == 2 ? 512u : in_flight == 3 ? 256u : 0u;
        if (max_nh >= 2 && (uint64_t)nr * max_nh <= limit) pair_stride = max_nh;
        // (the aligner raises "empty sequence" for every pair it meets; the chain only meets a kept read against its best allele.
        // With an empty haplotype or a read that is all soft clip in the batch the two ways could report differently for the
        // same input -- a read the filter drops, a haplotype that is nobody's best: such a call goes the chain's way.)
        for (uint32_t x = 0; x < nh && pair_stride; ++x)
            if (a.hap_off[x + 1] == a.hap_off[x]) pair_stride = 0;
        for (uint32_t r = 0; r < nr && pair_stride; ++r) {
            const uint32_t len = a.read_off[r + 1] - a.read_off[r];
            const uint32_t clipped = a.read_soft_clip ? a.read_soft_clip[2 * r] + a.read_soft_clip[2 * r + 1] : 0u;
            if (clipped >= len) pair_stride = 0;
        }
    }
    int pair_set = 0;  // 0 WHOLE: the handle's pair of the device's pool; 1 HALVES (see where the streams are used, below)
    if (pair_stride) {
        pair_set = (size_t)nr * pair_stride <= kHalvesUpToPairs ? 1 : 0;
        if (pair_set == 0 && !queues_acquire(h)) pair_set = 1;  // (the halves suit any small call)
        if (pair_set == 1 && !halves_acquire(h)) pair_stride = 0;  // (no queues to be had: the chain)
    }
    const Layout sizing(0, a, sw_capacity, pair_stride);
    phmm_batch *b = batch_create_in_arena(h, ng, a.region_read_off, a.region_hap_off, a.read_off, a.hap_off, a.out_off, sizing.end + 512);
    if (!b) return h->err_code ? h->err_code : PHMM_ERR_INVALID_ARG;
    const BatchView V = batch_view(b);
    Arena &A = h->A();
    hipStream_t S = h->S();  // (a call that aligns every pair moves to a stream of its own, below)
    Layout L(A.used, a, sw_capacity, pair_stride);
    int st = PHMM_OK;
    bool inline_rescue = false;
    auto bail = [&](int code) {
        std::string keep_err = h->err;
        (void)hipStreamSynchronize(S);
        for (hipStream_t t : h->swork.all_stream)
            if (t) (void)hipStreamSynchronize(t);
        phmm_batch_destroy(b);
        h->err = keep_err;
        return h->err_code = code;
    };
    if (L.end > A.cap) {  // cannot happen: `sizing` reserved for exactly this layout
        h->err = who + ": internal error, arena too small";
        return bail(PHMM_ERR_INTERNAL);
    }
    A.used = L.end;
    // ---- inputs into the pinned mirror --------------------------------------------------------------------------------
    auto put = [&](size_t at, const void *src, size_t bytes) {
        if (src && bytes) memcpy(A.host + at, src, bytes);
    };
    // (a combined flush of phmm_region_submit: every part's slice of an array, back to back, straight from its owner)
    auto put_each = [&](size_t at, auto ptr_of, auto bytes_of) {
        if (!parts) return put(at, ptr_of(a), bytes_of(a));
        size_t o = at;
        for (const RegionArgs &p : *parts) {
            put(o, ptr_of(p), bytes_of(p));
            o += bytes_of(p);
        }
    };
    auto n_r = [](const RegionArgs &p) { return (size_t)p.region_read_off[p.n_regions]; };
    auto n_h = [](const RegionArgs &p) { return (size_t)p.region_hap_off[p.n_regions]; };
    auto rbytes = [&](const RegionArgs &p) { return (size_t)p.read_off[n_r(p)]; };
    put_each(L.bases, [](const RegionArgs &p) { return (const void *)p.read_bases; }, rbytes);
    put_each(L.q0, [](const RegionArgs &p) { return (const void *)p.base_q; }, rbytes);
    put_each(L.i0, [](const RegionArgs &p) { return (const void *)p.ins_q; }, rbytes);
    put_each(L.d0, [](const RegionArgs &p) { return (const void *)p.del_q; }, rbytes);
    put_each(L.mapq, [](const RegionArgs &p) { return (const void *)p.mapq; }, n_r);
    put_each(L.haps, [](const RegionArgs &p) { return (const void *)p.hap_bases; }, [&](const RegionArgs &p) { return (size_t)p.hap_off[n_h(p)]; });
    put_each(L.refhap, [](const RegionArgs &p) { return (const void *)p.region_ref_hap; }, [](const RegionArgs &p) { return 4 * (size_t)p.n_regions; });
    put_each(L.pri, [](const RegionArgs &p) { return (const void *)p.hap_priority; }, [&](const RegionArgs &p) { return 4 * n_h(p); });
    put_each(L.rstart, [](const RegionArgs &p) { return (const void *)p.region_reference_start; }, [](const RegionArgs &p) { return 8 * (size_t)p.n_regions; });
    put_each(L.hc, [](const RegionArgs &p) { return (const void *)p.hap_cigar; }, [&](const RegionArgs &p) { return 4 * (size_t)p.hap_cigar_off[n_h(p)]; });
    put_each(L.hs, [](const RegionArgs &p) { return (const void *)p.hap_start_wrt_ref; }, [&](const RegionArgs &p) { return 4 * n_h(p); });
    put_each(L.oc, [](const RegionArgs &p) { return (const void *)p.orig_cigar; }, [&](const RegionArgs &p) { return 4 * (size_t)p.orig_cigar_off[n_r(p)]; });
    put_each(L.clip, [](const RegionArgs &p) { return (const void *)p.read_soft_clip; }, [&](const RegionArgs &p) { return 8 * n_r(p); });
    // (offset arrays: the combined ones)
    put(L.hco, a.hap_cigar_off, 4ull * (nh + 1));
    put(L.oco, a.orig_cigar_off, 4ull * (nr + 1));
    put(L.outco, a.out_cigar_off, 8ull * (nr + 1));
    memset(A.host + L.status_in, 0, 256);
    h->stat_staged_bytes += (2 + (a.ins_q ? 1 : 0) + (a.del_q ? 1 : 0)) * V.read_bytes + V.hap_bytes;
    // A small one-shot call (a region per call, the reference's pattern) does without the copy engine: a kernel fetches
    // the inputs from the pinned mirror, and the kernels store what the caller gets back straight into it.
    const size_t res_bytes = L.end - L.res;
    char *mirror = nullptr;
    // (also inside a combined flush of phmm_region_submit: this path has no copies to defer)
    if (zero_copy_allowed(h) && V.tight_out && nr && L.in_end <= stage_in_bytes() && res_bytes <= zero_copy_out_bytes()) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, A.host, 0) == hipSuccess && dp) mirror = (char *)dp;
    }
    char *const res_base = mirror ? mirror : A.dev;  // what only the caller reads
    if (!mirror) pair_stride = 0;  // (the aligner beside the other kernels reads its sequences from the mirror: small calls only)
    const size_t n_sw = pair_stride ? (size_t)nr * pair_stride : nr;  // alignments the aligner makes
    // ---- shapes ----------------------------------------------------------------------------------------------------------
    uint32_t max_r = 0, max_hap_cigar = 0;
    for (uint32_t r = 0; r < nr; ++r) max_r = std::max(max_r, a.read_off[r + 1] - a.read_off[r]);
    for (uint32_t x = 0; x < nh; ++x) max_hap_cigar = std::max(max_hap_cigar, a.hap_cigar_off[x + 1] - a.hap_cigar_off[x]);
    const bool align = nr && nh;  // (without haplotypes there is no best allele and nothing to align)
    SwGeometry G;
    size_t workers = 0;
    uint32_t pj_capacity = 0;
    if (align) {
        st = sw_plan(h, who, nr, V.max_h, std::max<uint32_t>(max_r, 1), &a.rcfg.sw_parameters, &G);
        if (st != PHMM_OK) return bail(st);
        if (G.ext_stride) pair_stride = 0;  // (refused below)
        workers = std::min<size_t>(G.max_workers, ((pair_stride ? n_sw : (size_t)nr) + G.gpb - 1) / G.gpb);
        pj_capacity = 4 * (sw_capacity + max_hap_cigar + 2) + 8;  // the lanes' builders: see phmm_cigar.cpp
        if (!ensure_sw_buffers(h, workers * G.slab_stride * 4, (size_t)nr * 4ull * pj_capacity * 4ull)) return bail(PHMM_ERR_HIP);
    } else {
        pair_stride = 0;
    }
    phmm_handle::SwWork &W = h->swork;
    hipStream_t T_all = nullptr;  // the all-pairs aligner's stream
    if (pair_stride) {
        // Two streams with hardware queues of their own, created one behind the other: the call's kernels run on the first,
        // the all-pairs aligner on the second.  What was measured (rocprofv3 queue ids, 128 x 8 reads x haplotypes per call):
        // the slot stream and an ordinary second stream may share one of the runtime's four queues (one kernel after the
        // other: 166 us per call); two queues whose ids are equal modulo four -- one pipe of the command processor --
        // stretch the pre-step from 20 to 57 us (160 us per call); a stream of another PRIORITY delivers its event ~55 us
        // late (190 us); queues with consecutive ids: 120 us.  A stream with a CU mask always gets a new queue.
        // Two such pairs.  WHOLE: both masks name every CU -- a call whose kernels fill the chip anyway (128 x 8: a wave per
        // SIMD each).  HALVES: the call's kernels on one half of the CUs, the aligner on the other -- a call of a few hundred
        // pairs, whose waves the dispatcher would otherwise put on the SAME first CUs of every XCD although nine tenths of the
        // chip are idle (30 x 3: PairHMM kernel 59 us beside the aligner on shared CUs, 37 on CUs of its own, as alone).
        const int set = pair_set;  // (chosen -- and its queues made sure of -- where the call decided to align every pair)
        if (W.pair_main[set]) S = W.pair_main[set];
        T_all = W.all_stream[set];
    }
    // (the last kernel of a small call reports to the calling thread through the mirror: region_finish)
    const bool flag_wait = mirror && !chained && nr && h->sw.region_flag_wait;
    if ((pair_stride || flag_wait) && !W.d_pair_done) {
        if (!ok(h, hipMalloc((void **)&W.d_pair_done, 256), "hipMalloc") || !ok(h, hipMemset(W.d_pair_done, 0, 256), "hipMemset")) return bail(PHMM_ERR_HIP);
        W.pair_done_target = 0;
        W.finish_count = 0;
    }
    bool good;
    if (mirror) {  // (the inputs are fetched by blocks of the pre-step's launch, below)
        canary_staged(h, A, L.in_end);
        memset(A.host + L.res, 0, 256);
        batch_set_status(b, (uint32_t *)(A.dev + L.status_in));
        good = true;
    } else {
        batch_set_status(b, (uint32_t *)(A.dev + L.res));
        good = ok(h, hipMemcpyAsync(A.dev, A.host, L.in_end, hipMemcpyHostToDevice, S), "H2D batch") &&
               ok(h, hipMemsetAsync(A.dev + L.res, 0, 256, S), "memset status");
    }
    // ---- the aligner, every read against every haplotype of its region: beside everything below, straight from the mirror ------
    auto sw_params = [&](const char *in_base) {
        SwParams sp{};
        sp.a_begin = 0;
        sp.n_alignments = (uint32_t)n_sw;
        sp.ref_off = (const uint32_t *)(in_base + ((const char *)V.d_hap_off - A.dev));
        sp.alt_off = (const uint32_t *)(in_base + ((const char *)V.d_read_off - A.dev));
        sp.ref_index = (const uint32_t *)(A.dev + L.refidx);
        sp.ref_bases = (const uint8_t *)(in_base + L.haps);
        sp.alt_bases = (const uint8_t *)(in_base + L.bases);
        sp.w_match = a.rcfg.sw_parameters.match_value;
        sp.w_mismatch = a.rcfg.sw_parameters.mismatch_penalty;
        sp.w_open = a.rcfg.sw_parameters.gap_open_penalty;
        sp.w_extend = a.rcfg.sw_parameters.gap_extend_penalty;
        sp.strategy = a.rcfg.overhang_strategy;
        sp.cigar_off = nullptr;
        sp.cigar_slot = sw_capacity;
        sp.alt_clip = a.read_soft_clip ? (const uint32_t *)(in_base + L.clip) : nullptr;
        sp.cigar = (uint32_t *)(A.dev + L.swc);
        sp.n_cigar = (uint32_t *)(A.dev + L.nsw);
        sp.alignment_offset = (int32_t *)(A.dev + L.swo);
        sp.slab = W.slab;
        sp.slab_stride = G.slab_stride;
        sp.status = (uint32_t *)(res_base + L.res + 64);
        sp.max_ref = V.max_h;
        sp.max_alt = std::max<uint32_t>(max_r, 1);
        sp.lds_ref_bytes = (uint32_t)G.lds_ref;
        sp.lds_alt_bytes = (uint32_t)G.lds_alt;
        sp.lds_group_bytes = (uint32_t)G.lds_group;
        sp.groups_per_block = (uint32_t)G.gpb;
        return sp;
    };
    // (enqueued FIRST: started 10 us ahead of the pre-step it runs 80 us beside pre-step + PairHMM's 20 + 64 -- the waves of the
    // two kernels share the SIMDs' issue slots, alone they take 57 and 15 + 40; started together with the PairHMM kernel it
    // takes 94 and that one 85.  Raising either kernel's wave priority (switch region_prio) only moves the time to the other.)
    auto launch_all_pairs = [&]() {
        SwParams sp = sw_params(mirror);
        sp.ref_index = nullptr;
        sp.pair_stride = pair_stride;
        sp.pair_single_nh = ng == 1 ? nh : 0u;
        sp.report_clock = (h->sw.region_debug_pick & 2) ? 2u : 0u;  // (never the clock words here; 2 = the canary's negative control)
        sp.read_region = (const uint32_t *)(mirror + ((const char *)V.d_read_region - A.dev));
        sp.region_hap_off = (const uint32_t *)(mirror + ((const char *)V.d_region_hap_off - A.dev));
        // (no event between the streams: the aligner's blocks count themselves in, and the kernel that consumes the alignments
        // -- enqueued right behind the PairHMM kernel -- waits for the count: an event that is still pending reaches the other
        // queue ~12 us late, one that has fired still costs ~6 us of barrier packet)
        sp.done_counter = W.d_pair_done;
        good = ok(h, launch_sw(G.L, G.K, G.transposed, G.variant, sp, (uint32_t)workers, G.lds, T_all), "phmm_sw_align_kernel (all pairs)");
        if (good) {
            W.pair_done_target += (uint32_t)workers;  // (only what was launched is waited for)
            W.region_sw_all_calls += 1;
        }
    };
    // (tests: the aligner BEHIND phmm_pick_reads on the call's own stream -- in order on one queue the wait cannot be met)
    const bool all_pairs_behind_pick = pair_stride && (h->sw.region_debug_pick & 1) != 0;
    if (all_pairs_behind_pick) T_all = S;
    if (good && pair_stride && !all_pairs_behind_pick) launch_all_pairs();
    // ---- pre-step ----------------------------------------------------------------------------------------------------------
    if (good && nr) {
        PrepParams pp{};
        pp.n_reads = nr;
        // a small call: the pre-step reads the pinned mirror itself, and its launch carries the copy of everything staged
        char *const in_base = mirror ? mirror : A.dev;
        pp.read_off = mirror ? (const uint32_t *)(mirror + ((const char *)V.d_read_off - A.dev)) : V.d_read_off;
        pp.read_bases = (const uint8_t *)(in_base + L.bases);
        pp.base_q = (const uint8_t *)(in_base + L.q0);
        pp.ins_q = a.ins_q ? (const uint8_t *)(in_base + L.i0) : nullptr;
        pp.del_q = a.del_q ? (const uint8_t *)(in_base + L.d0) : nullptr;
        pp.mapq = (const uint8_t *)(in_base + L.mapq);
        pp.stage_src = mirror;
        pp.stage_dst = A.dev;
        pp.stage_n16 = mirror ? (uint32_t)((L.in_end + 15) / 16) : 0u;
        pp.pcr_cache = a.cfg.pcr_error_model ? h->d_pcr_cache + 128 * a.cfg.pcr_error_model : nullptr;
        pp.out_q = (uint8_t *)(A.dev + L.q);
        pp.out_ins = (uint8_t *)(A.dev + L.i);
        pp.out_del = (uint8_t *)(A.dev + L.d);
        pp.out_gcp = (uint8_t *)(A.dev + L.g);
        pp.threshold = (double *)(A.dev + L.thr);
        pp.lds_rows = (uint32_t)((max_r + 1 + 7) / 8 * 8);
        pp.waves_per_read = nr <= 2048 ? std::max<uint32_t>(1, (max_r + 63) / 64) : 1;  // few reads: a wave per 64 positions
        pp.default_indel_qual = 45;  // ReadUtils::DEFAULT_INSERTION_DELETION_QUAL (read_utils.rs:23)
        pp.constant_gcp = a.cfg.constant_gcp;
        pp.base_quality_score_threshold = a.cfg.base_quality_score_threshold;
        pp.disable_cap_to_mapq = a.cfg.disable_cap_read_qualities_to_mapq;
        pp.dynamic_disqualification = a.cfg.dynamic_read_disqualification;
        pp.read_disqualification_scale = a.cfg.read_disqualification_scale;
        pp.expected_error_rate_per_base = a.cfg.expected_error_rate_per_base;
        if ((size_t)pp.lds_rows * 17 * 4 > 160 * 1024) {
            h->err = who + ": read too long for the pre-step kernel";
            return bail(PHMM_ERR_INVALID_ARG);
        }
        good = ok(h, launch_prep(pp, S), "phmm_prep_reads");
    }
    // ---- PairHMM ---------------------------------------------------------------------------------------------------------
    // The exact pass below -600 rides in-stream -- unless no pair of this batch can get there: every likelihood is at least
    // the path "first base matched anywhere, everything else inserted", 10^-(q/10)/3 x (1 - 10^-(gcp/10)) x 10^-(ins/10) x
    // 10^-(gcp/10) per further base, i.e. above -(25.5 + 0.5) - 0.7 - 25.5 - (R - 2) gcp / 10 for u8 qualities, and the
    // pre-step sets gcp to the engine's constant (10 -> reads shorter than 540 bases never need the pass).
    if (good && nr) {
        const bool can_underflow = a.cfg.constant_gcp == 0 || 53.0 + (double)max_r * a.cfg.constant_gcp / 10.0 >= 590.0;
        if (can_underflow) good = batch_set_inline_rescue(h, b);
        inline_rescue = can_underflow && !h->sw.no_rescue;
        if (!good) return bail(PHMM_ERR_HIP);
        good = phmm_batch_bind_device(b, (const uint8_t *)(A.dev + L.bases), (const uint8_t *)(A.dev + L.q), (const uint8_t *)(A.dev + L.i),
                                      (const uint8_t *)(A.dev + L.d), (const uint8_t *)(A.dev + L.g), (const uint8_t *)(A.dev + L.haps),
                                      (double *)(A.dev + L.out)) == PHMM_OK &&
               phmm_batch_launch(b, S) == PHMM_OK;
    }
    // ---- post-step + best alleles: one kernel, the matrix stays where the forward kernels wrote it ---------------------------
    PostBestParams pb{};
    if (good && nr) {
        PostParams &po = pb.post;
        po.n_reads = nr;
        po.read_region = V.d_read_region;
        po.region_read_off = V.d_region_read_off;
        po.region_hap_off = V.d_region_hap_off;
        po.out_off = V.d_out_off;
        po.region_ref_hap = (const int32_t *)(A.dev + L.refhap);
        po.out = (double *)(A.dev + L.out);
        po.out_final = mirror ? (double *)(mirror + L.out) : nullptr;
        po.threshold = (const double *)(A.dev + L.thr);
        po.keep = (uint8_t *)(A.dev + L.keep);
        po.status_in = mirror ? (const uint32_t *)(A.dev + L.status_in) : nullptr;
        po.status_out = mirror ? (uint32_t *)(mirror + L.res) : nullptr;
        po.max_likelihood_difference_cap = a.cfg.log10_global_read_mismapping_rate;
        po.symmetric = a.cfg.symmetrically_normalize_alleles_to_reference;
        BestParams &bp = pb.best;
        bp.r_begin = 0;
        bp.n_reads = nr;
        bp.n_regions = ng;
        bp.region_read_off = V.d_region_read_off;
        bp.region_hap_off = V.d_region_hap_off;
        bp.out_off = V.d_out_off;
        bp.likelihoods = (const double *)(A.dev + L.out);
        bp.keep = (const uint8_t *)(A.dev + L.keep);
        bp.priority = a.hap_priority ? (const int32_t *)(A.dev + L.pri) : nullptr;
        bp.threshold = a.rcfg.informative_threshold;
        bp.best_allele = (int32_t *)(res_base + L.best);
        bp.likelihood = (double *)(res_base + L.lk);
        bp.confidence = (double *)(res_base + L.conf);
        bp.ref_index = (uint32_t *)(A.dev + L.refidx);
        pb.skip_single_allele = (a.rcfg.flags & PHMM_REGION_SKIP_SINGLE_ALLELE) ? 1u : 0u;
        pb.keep_final = mirror ? (uint8_t *)(mirror + L.keep) : nullptr;
        if (!pair_stride) good = ok(h, launch_post_best(pb, S), "phmm_post_best_reads");  // (else: part of phmm_pick_reads, below)
    }
    // ---- alignments to the best haplotypes and their projection onto the reference -----------------------------------------
    ProjectParams pj{};
    if (nr) {
        pj.r_begin = 0;
        pj.n_reads = nr;
        pj.n_regions = ng;
        pj.region_read_off = V.d_region_read_off;
        pj.region_hap_off = V.d_region_hap_off;
        pj.read_off = V.d_read_off;
        pj.read_bases = (const uint8_t *)(A.dev + L.bases);
        pj.hap_off = V.d_hap_off;
        pj.hap_bases = (const uint8_t *)(A.dev + L.haps);
        pj.region_ref_hap = (const int32_t *)(A.dev + L.refhap);
        pj.region_reference_start = (const uint64_t *)(A.dev + L.rstart);
        pj.hap_cigar_off = (const uint32_t *)(A.dev + L.hco);
        pj.hap_cigar = (const uint32_t *)(A.dev + L.hc);
        pj.hap_start_wrt_ref = (const uint32_t *)(A.dev + L.hs);
        pj.best_allele = nullptr;  // (derived from ref_index, which lives on the device)
        pj.ref_index = (const uint32_t *)(A.dev + L.refidx);
        pj.sw_cigar_off = nullptr;
        pj.sw_cigar_slot = sw_capacity;
        pj.sw_pair_stride = pair_stride;
        pj.sw_cigar = (const uint32_t *)(A.dev + L.swc);
        pj.n_sw_cigar = (const uint32_t *)(A.dev + L.nsw);
        pj.sw_offset = (const int32_t *)(A.dev + L.swo);
        pj.read_clip = a.read_soft_clip ? (const uint32_t *)(A.dev + L.clip) : nullptr;
        pj.orig_cigar_off = (const uint32_t *)(A.dev + L.oco);
        pj.orig_cigar = (const uint32_t *)(A.dev + L.oc);
        pj.out_cigar_off = (const uint64_t *)(A.dev + L.outco);
        pj.out_cigar = (uint32_t *)(res_base + L.pout);
        pj.n_out_cigar = (uint32_t *)(res_base + L.pno);
        pj.new_pos = (int64_t *)(res_base + L.pos);
        pj.status = (int32_t *)(res_base + L.pst);
        pj.flags = (uint32_t *)(res_base + L.res + 128);
        pj.workspace = W.ws;
        pj.capacity = pj_capacity;
        if (flag_wait) {
            pj.finish_counter = W.d_pair_done + 16;
            pj.finish_target = W.finish_count;  // (the launch adds its blocks)
            pj.finish_flag = (uint32_t *)(mirror + L.res + 224);
        }
    }
    bool used_lite = false;
    if (good && align && G.ext_stride) {  // (reads and haplotypes never get there; the aligner's own entry points handle such lengths)
        h->err = who + ": sequences too long for the per-region pipeline (about 8 000 bases each)";
        return bail(PHMM_ERR_INVALID_ARG);
    }
    if (good && pair_stride) {  // the alignments were made beside all this: post-step + best allele, wait for them, projection
        pj.wait_counter = W.d_pair_done;
        pj.wait_target = all_pairs_behind_pick ? W.pair_done_target + (uint32_t)workers : W.pair_done_target;
        pj.wait_ticks = (uint32_t)std::min<uint64_t>(100ull * (uint64_t)std::max(1, h->sw.region_pick_timeout_us), 0xffffffffull);
        uint32_t blocks = 0;
        good = ok(h, launch_pick(pb, pj, S, &blocks), "phmm_pick_reads");
        if (good && pj.finish_counter) W.finish_count += blocks;  // (only blocks that were launched count themselves in)
        if (good && all_pairs_behind_pick) launch_all_pairs();
    } else if (good && align) {
        SwParams sp = sw_params(A.dev);
        // (chunks of one call follow each other through the handle's one slab and workspace)
        if (chained && W.region_sw_pending) good = ok(h, hipStreamWaitEvent(S, W.region_sw_done, 0), "hipStreamWaitEvent");
        // (reads against their haplotypes: the tags-only sweep first, the full instance over the alignments that met a gap;
        // the counter is a word of the status block that is staged as zeros with the inputs)
        // (a call that met gaps in more than three alignments of ten sends the handle's next fifteen straight to the full instance)
        bool lite = G.variant == SW_PLAIN && h->sw.sw_lite != 0 && (sp.strategy == PHMM_SW_SOFTCLIP || sp.strategy == PHMM_SW_IGNORE);
        if (lite && h->sw.sw_lite < 0 && W.lite_skip > 0) {
            W.lite_skip -= 1;
            lite = false;
        }
        used_lite = lite;
        if (lite) {
            SwParams s1 = sp, s2 = sp;
            s1.todo_out = (uint32_t *)(A.dev + L.todo);
            s1.todo_out_count = (uint32_t *)(A.dev + L.status_in) + 32;
            s2.todo = s1.todo_out;
            s2.todo_count = s1.todo_out_count;
            s2.feedback = (uint32_t *)(res_base + L.res + 192);
            good = good && ok(h, launch_sw(G.L, G.K, G.transposed, SW_LITE, s1, (uint32_t)workers, G.lds, S), "phmm_sw_align_kernel (tags)") &&
                   ok(h, launch_sw(G.L, G.K, G.transposed, G.variant, s2, (uint32_t)workers, G.lds, S), "phmm_sw_align_kernel");
        } else {
            good = good && ok(h, launch_sw(G.L, G.K, G.transposed, G.variant, sp, (uint32_t)workers, G.lds, S), "phmm_sw_align_kernel");
        }
    } else if (good && nr) {  // nothing was aligned: the kernels behind the aligner still find defined alignments
        // (the projection below works in the handle's one workspace like every chunk's: behind the chunk before it)
        if (chained && W.region_sw_pending) good = ok(h, hipStreamWaitEvent(S, W.region_sw_done, 0), "hipStreamWaitEvent");
        good = good && ok(h, hipMemsetAsync(A.dev + L.nsw, 0, 4ull * nr, S), "memset") && ok(h, hipMemsetAsync(A.dev + L.swo, 0, 4ull * nr, S), "memset");
    }
    if (good && nr && !pair_stride) {
        if (!align && !ensure_sw_buffers(h, 0, (size_t)nr * 4ull * (4 * (sw_capacity + max_hap_cigar + 2) + 8) * 4ull)) return bail(PHMM_ERR_HIP);
        if (!align) {
            pj.workspace = W.ws;
            pj.capacity = 4 * (sw_capacity + max_hap_cigar + 2) + 8;
        }
        uint32_t blocks = 0;
        good = ok(h, launch_project(pj, S, &blocks), "phmm_project_kernel");
        if (good && pj.finish_counter) W.finish_count += blocks;  // (only blocks that were launched count themselves in)
        if (good && chained) {
            if (!W.region_sw_done) good = ok(h, hipEventCreateWithFlags(&W.region_sw_done, hipEventDisableTiming), "hipEventCreate");
            good = good && ok(h, hipEventRecord(W.region_sw_done, S), "hipEventRecord");
            W.region_sw_pending = good;
        }
    }
    const bool eager = eager_d2h(h);  // otherwise region_finish fetches the results
    if (good && eager && !mirror)
        good = ok(h, hipMemcpyAsync(A.host + L.res, A.dev + L.res, res_bytes, hipMemcpyDeviceToHost, S), "D2H results");
    if (!good) return bail(h->err_code ? h->err_code : PHMM_ERR_HIP);
    pending->b = b;
    pending->slot = h->slot;
    pending->stream = S;
    pending->a = a;
    pending->parts = parts;
    pending->L = L;
    pending->d2h_pending = !eager && !mirror;
    pending->zero_copy = mirror != nullptr;
    pending->sw_capacity = sw_capacity;
    pending->pair_stride = pair_stride;
    pending->finish_flag = flag_wait ? (const uint32_t *)(A.host + L.res + 224) : nullptr;
    pending->all_stream = pair_stride ? T_all : nullptr;
    pending->inline_rescue = inline_rescue;
    pending->lite = used_lite;
    return PHMM_OK;
}

// Wait for a pending batch and hand everything to the caller.  *sw_needed > 0: a read -> haplotype alignment outgrew its
// slot of `sw_capacity` elements (the caller runs the batch again with larger ones); nothing was handed over then.
int region_finish(phmm_handle *h, PendingRegion *p, uint32_t *sw_needed) {
    if (sw_needed) *sw_needed = 0;
    if (!p->b) return PHMM_OK;
    phmm_batch *b = p->b;
    const RegionArgs &a = p->a;
    const Layout &L = p->L;
    const Arena &A = h->arenas[p->slot];
    hipStream_t S = p->stream ? p->stream : h->streams[p->slot];
    const uint32_t ng = a.n_regions, nr = a.region_read_off[ng];
    int st = PHMM_OK;
    auto done = [&](int code) {
        std::string keep_err = h->err;
        phmm_batch_destroy(b);
        h->err = keep_err;
        p->b = nullptr;
        if (code != PHMM_OK && code != kPickTimedOut) h->err_code = code;
        return code;
    };
    // A small call's last kernel has told this thread itself, through the mirror, when its last block was through: the runtime
    // reports the same ~6 us later (tools/ubench/sync_latency.hip).  The stream is left as it is -- in order, and the next
    // call's kernels queue up behind a kernel that has nothing left to do.  (No word within 2 ms: the ordinary wait, which
    // also surfaces a fault.)
    bool told = false;
    if (p->finish_flag) {
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        const bool naps = more_callers_than_cores(h);  // (spinning waiters beyond the process' cores starve the callers that stage)
        for (uint32_t spins = 0; !told; ++spins) {
            told = __atomic_load_n(p->finish_flag, __ATOMIC_ACQUIRE) != 0;
            if (!told && (spins & 255u) == 255u && std::chrono::steady_clock::now() >= give_up) break;
            if (!told && naps && (spins & 63u) == 63u) std::this_thread::sleep_for(std::chrono::microseconds(20));
            if (!told) __builtin_ia32_pause();
        }
    }
    if ((!told && !ok(h, wait_stream(h, S), "sync")) ||
        (p->d2h_pending && (!ok(h, hipMemcpyAsync(A.host + L.res, A.dev + L.res, L.end - L.res, hipMemcpyDeviceToHost, S), "D2H results") ||
                            !ok(h, wait_stream(h, S), "sync(D2H)"))))
        return done(PHMM_ERR_HIP);
    const char *hs = A.host;
    const uint32_t *sw_st = (const uint32_t *)(hs + L.res + 64);
    if (p->pair_stride && ((const uint32_t *)(hs + L.res + 128))[1]) {
        // phmm_pick_reads ran out of time waiting for the aligner on the other stream (the two were not running side by side: one
        // hardware queue for both, an unmapped queue, a stalled device).  Let the aligner finish -- it writes into this call's
        // arena -- and hand nothing over: region_one_shot runs the call again as the chain, which needs no second queue.
        h->swork.region_pick_timeouts += 1;
        const bool synced = ok(h, hipStreamSynchronize(S), "sync") && (!p->all_stream || ok(h, hipStreamSynchronize(p->all_stream), "sync(all pairs)"));
        (void)canary_after_call(h, h->arenas[p->slot], L.res, L.end - L.res);
        return done(synced ? kPickTimedOut : PHMM_ERR_HIP);
    }
    if (p->lite) {
        const uint64_t again = *(const uint32_t *)(hs + L.res + 192);
        h->swork.last_second_pass = again;
        if (again * 10 > (uint64_t)nr * 3) h->swork.lite_skip = 15;
    } else {
        h->swork.last_second_pass = 0;
    }
    if (nr && sw_st[SW_STATUS_CAPACITY]) {  // (with every pair aligned: whichever pair's; the call goes round again the plain way)
        std::vector<uint32_t> n_sw(p->pair_stride ? (size_t)nr * p->pair_stride : nr);
        if (!ok(h, hipMemcpy(n_sw.data(), A.dev + L.nsw, 4ull * n_sw.size(), hipMemcpyDeviceToHost), "D2H sw")) return done(PHMM_ERR_HIP);
        if (sw_needed) *sw_needed = *std::max_element(n_sw.begin(), n_sw.end());
        h->err = "phmm_region_compute: a read -> haplotype CIGAR needs more elements than the library reserved";
        return done(PHMM_ERR_CIGAR_CAPACITY);
    }
    if (nr && !p->parts) {
        memcpy(a.keep, hs + L.keep, nr);
        batch_copy_out(b, (const double *)(hs + L.out), a.out);
        memcpy(a.best_allele, hs + L.best, 4ull * nr);
        memcpy(a.likelihood, hs + L.lk, 8ull * nr);
        memcpy(a.confidence, hs + L.conf, 8ull * nr);
        memcpy(a.status, hs + L.pst, 4ull * nr);
        memcpy(a.n_out_cigar, hs + L.pno, 4ull * nr);
        memcpy(a.new_pos, hs + L.pos, 8ull * nr);
        if (a.out_cigar_off[nr]) memcpy(a.out_cigar, hs + L.pout, 4ull * a.out_cigar_off[nr]);
    } else if (nr) {  // every part gets its reads' and regions' share (its own out_off may leave gaps: region by region)
        size_t r0 = 0, g0 = 0;
        const double *v = (const double *)(hs + L.out);
        for (const RegionArgs &q : *p->parts) {
            const size_t nrp = q.region_read_off[q.n_regions];
            if (nrp) {
                memcpy(q.keep, hs + L.keep + r0, nrp);
                memcpy(q.best_allele, hs + L.best + 4 * r0, 4 * nrp);
                memcpy(q.likelihood, hs + L.lk + 8 * r0, 8 * nrp);
                memcpy(q.confidence, hs + L.conf + 8 * r0, 8 * nrp);
                memcpy(q.status, hs + L.pst + 4 * r0, 4 * nrp);
                memcpy(q.n_out_cigar, hs + L.pno + 4 * r0, 4 * nrp);
                memcpy(q.new_pos, hs + L.pos + 8 * r0, 8 * nrp);
                if (q.out_cigar_off[nrp]) memcpy(q.out_cigar, hs + L.pout + 4 * a.out_cigar_off[r0], 4ull * q.out_cigar_off[nrp]);
            }
            for (uint32_t g = 0; g < q.n_regions; ++g) {
                const uint64_t cnt = (uint64_t)(q.region_read_off[g + 1] - q.region_read_off[g]) * (q.region_hap_off[g + 1] - q.region_hap_off[g]);
                if (cnt) memcpy(q.out + q.out_off[g], v + a.out_off[g0 + g], 8 * cnt);
            }
            r0 += nrp;
            g0 += q.n_regions;
        }
    }
    // (with the exact pass in-stream its verdict counts: a fast kernel may have raised the bit for a pair the pass replaced)
    if (status_positive(*(const uint32_t *)(hs + L.res), p->inline_rescue)) {
        h->err = "PairHmm Log Probability cannot be greater than 0.0";  // pair_hmm.rs:478-481
        st = PHMM_ERR_POSITIVE_RESULT;
    } else if (nr && sw_st[SW_STATUS_EMPTY]) {  // the reference asserts (smith_waterman_aligner.rs:65-68, :132-134)
        h->err = "phmm_region_compute: non-empty sequences are required for the Smith-Waterman calculation";
        st = PHMM_ERR_INVALID_ARG;
    } else if (nr && (*(const uint32_t *)(hs + L.res + 128) & 1u)) {
        h->err = "phmm_region_compute: a CIGAR needs more elements than its slot holds (n_out_cigar has the sizes)";
        st = PHMM_ERR_CIGAR_CAPACITY;
    }
    // (PHMM_MIRROR_CANARY: the results are with the caller -- nothing may store into this call's block from here on)
    if (p->zero_copy && !canary_after_call(h, h->arenas[p->slot], L.res, L.end - L.res) && st == PHMM_OK) st = PHMM_ERR_INTERNAL;
    return done(st);
}

// one batch, start to end, on the current slot; grows the alignments' slots once if one of them needs it
int region_one_shot(phmm_handle *h, const RegionArgs &a, const std::vector<RegionArgs> *parts, uint32_t *sw_capacity) {
    const InFlight in_flight(h->device);
    bool may_align_all = true, grown = false;
    for (;;) {
        PendingRegion p;
        int st = region_enqueue(h, a, parts, *sw_capacity, false, may_align_all, &p);
        uint32_t needed = 0;
        if (st == PHMM_OK) st = region_finish(h, &p, &needed);
        if (st == PHMM_ERR_CIGAR_CAPACITY && needed > *sw_capacity && !grown) {
            *sw_capacity = needed;
            grown = true;
            may_align_all = false;
            continue;
        }
        if (st == kPickTimedOut) {  // (only a call that aligned every pair comes back with this)
            if (may_align_all) {
                may_align_all = false;
                continue;
            }
            h->err = "phmm_region_compute: internal error, the chained call waited for an all-pairs aligner";
            return h->err_code = PHMM_ERR_INTERNAL;
        }
        return st;
    }
}

// The caller's arguments restricted to the regions of one chunk: offsets rebased to zero, every pointer moved on.
struct ChunkArgs {
    RegionArgs a;
    std::vector<uint32_t> hco, oco;
    std::vector<uint64_t> outco;
    void build(const RegionArgs &w, const ChunkView &c) {
        a = w;
        a.n_regions = c.g1 - c.g0;
        a.region_read_off = c.rro.data();
        a.region_hap_off = c.rho.data();
        a.read_off = c.ro.data();
        a.hap_off = c.ho.data();
        a.out_off = c.oo.data();
        const size_t bo = c.read_byte0, co = c.hap_byte0;
        auto mv = [](auto *p, size_t by) { return p ? p + by : p; };
        a.read_bases = mv(w.read_bases, bo);
        a.base_q = mv(w.base_q, bo);
        a.ins_q = mv(w.ins_q, bo);
        a.del_q = mv(w.del_q, bo);
        a.mapq = mv(w.mapq, c.r0);
        a.read_soft_clip = mv(w.read_soft_clip, 2ull * c.r0);
        a.hap_bases = mv(w.hap_bases, co);
        a.region_ref_hap = mv(w.region_ref_hap, c.g0);
        a.hap_priority = mv(w.hap_priority, c.h0);
        a.region_reference_start = mv(w.region_reference_start, c.g0);
        a.hap_start_wrt_ref = mv(w.hap_start_wrt_ref, c.h0);
        hco.resize(c.h1 - c.h0 + 1);
        for (uint32_t x = c.h0; x <= c.h1; ++x) hco[x - c.h0] = w.hap_cigar_off[x] - w.hap_cigar_off[c.h0];
        oco.resize(c.r1 - c.r0 + 1);
        outco.resize(c.r1 - c.r0 + 1);
        for (uint32_t r = c.r0; r <= c.r1; ++r) {
            oco[r - c.r0] = w.orig_cigar_off[r] - w.orig_cigar_off[c.r0];
            outco[r - c.r0] = w.out_cigar_off[r] - w.out_cigar_off[c.r0];
        }
        a.hap_cigar_off = hco.data();
        a.orig_cigar_off = oco.data();
        a.out_cigar_off = outco.data();
        a.hap_cigar = mv(w.hap_cigar, w.hap_cigar_off[c.h0]);
        a.orig_cigar = mv(w.orig_cigar, w.orig_cigar_off[c.r0]);
        a.out = mv(w.out, w.out_off[c.g0]);
        a.keep = mv(w.keep, c.r0);
        a.best_allele = mv(w.best_allele, c.r0);
        a.likelihood = mv(w.likelihood, c.r0);
        a.confidence = mv(w.confidence, c.r0);
        a.out_cigar = mv(w.out_cigar, w.out_cigar_off[c.r0]);
        a.n_out_cigar = mv(w.n_out_cigar, c.r0);
        a.new_pos = mv(w.new_pos, c.r0);
        a.status = mv(w.status, c.r0);
    }
};

}  // namespace

namespace phmm_host {

// the regions of several submissions as ONE batch on `h` (a combined flush of phmm_region_submit): `combined` holds the
// concatenated offset arrays and the shared configuration, payload and results are the parts'
int region_compute_parts(phmm_handle *h, const RegionArgs &combined, const std::vector<RegionArgs> &parts) {
    DevGuard dg(h->device);
    if (!combined.region_read_off[combined.n_regions]) return PHMM_OK;
    uint32_t sw_capacity = kFirstSwCapacity;
    h->slot = 0;
    return region_one_shot(h, combined, &parts, &sw_capacity);
}

int region_compute(phmm_handle *h, const RegionArgs &a) {
    DevGuard dg(h->device);
    const uint32_t ng = a.n_regions, nr = a.region_read_off[ng];
    if (!nr) return PHMM_OK;  // no reads: no likelihoods, nothing to realign
    uint32_t sw_capacity = kFirstSwCapacity;
    // ---- small / medium batch: one shot -----------------------------------------------------------------------------------
    if (ng < 8 || (size_t)a.read_off[nr] <= one_shot_bytes() || h->sw.no_pipeline) {
        h->slot = 0;
        return region_one_shot(h, a, nullptr, &sw_capacity);
    }
    // ---- large batch: chunks of regions through the kSlots (arena, stream) pairs, like phmm_engine_compute: staging and
    //      the H2D copy of chunk i+1 overlap the kernels of chunk i.  Every region is independent, a chunk only rebases offsets.
    struct Slot {
        PendingRegion pend;
        ChunkArgs args;
    } slots[kSlots];
    struct Drain {  // an exception on the way (host allocation) must not leave chunks in flight
        phmm_handle *h;
        Slot *s;
        ~Drain() {
            for (int i = 0; i < kSlots; ++i)
                if (s[i].pend.b) {
                    (void)hipStreamSynchronize(h->streams[s[i].pend.slot]);
                    phmm_batch_destroy(s[i].pend.b);
                    s[i].pend.b = nullptr;
                }
            h->slot = 0;
            h->defer_d2h = false;
            h->swork.region_sw_pending = false;
        }
    } drain{h, slots};
    int st = PHMM_OK, final_st = PHMM_OK;
    // a chunk's own verdict: what the whole call reports is the first failure, but an output slot that is too small
    // (PHMM_ERR_CIGAR_CAPACITY of the projection) does not stop the other chunks -- n_out_cigar has to be complete
    auto finish_slot = [&](Slot &sl) {
        if (!sl.pend.b) return PHMM_OK;
        uint32_t needed = 0;
        const int slot = sl.pend.slot;
        int s2 = region_finish(h, &sl.pend, &needed);
        if (s2 == PHMM_ERR_CIGAR_CAPACITY && needed > sw_capacity) {  // this chunk once more, alone, with larger slots
            sw_capacity = needed;
            const int keep_slot = h->slot;
            h->slot = slot;
            h->defer_d2h = false;
            for (int i = 0; i < kSlots; ++i) (void)hipStreamSynchronize(h->streams[i]);  // (the slab and the workspace are shared)
            h->swork.region_sw_pending = false;
            s2 = region_one_shot(h, sl.args.a, nullptr, &sw_capacity);
            h->defer_d2h = true;
            h->slot = keep_slot;
        }
        if (s2 == PHMM_ERR_CIGAR_CAPACITY) {
            if (final_st == PHMM_OK) final_st = s2;
            return PHMM_OK;
        }
        return s2;
    };
    ChunkView c;
    c.mixed = true;  // (chunks large enough for the chained kernel whatever the shapes)
    {
        const uint32_t nr0 = a.region_read_off[1] - a.region_read_off[0], nh0 = a.region_hap_off[1] - a.region_hap_off[0];
        const uint32_t hl0 = nh0 ? a.hap_off[1] - a.hap_off[0] : 0;
        bool mixed = false;
        for (uint32_t g = 1; g < ng && !mixed; ++g)
            mixed = a.region_read_off[g + 1] - a.region_read_off[g] != nr0 || a.region_hap_off[g + 1] - a.region_hap_off[g] != nh0 ||
                    (nh0 && a.hap_off[a.region_hap_off[g] + 1] - a.hap_off[a.region_hap_off[g]] != hl0);
        c.mixed = mixed;
    }
    int n_chunks = 0;
    h->defer_d2h = true;
    h->swork.region_sw_pending = false;
    while (st == PHMM_OK && next_chunk(c, ng, a.region_read_off, a.region_hap_off, a.read_off, a.hap_off, a.out_off)) {
        const int slot = n_chunks % kSlots;
        st = finish_slot(slots[slot]);  // the slot's previous chunk must be out of its arena
        if (st != PHMM_OK) break;
        h->slot = slot;
        slots[slot].args.build(a, c);
        st = region_enqueue(h, slots[slot].args.a, nullptr, sw_capacity, true, false, &slots[slot].pend);
        ++n_chunks;
    }
    for (int i = 0; i < kSlots; ++i) {  // drain in submission order
        const int s2 = finish_slot(slots[(n_chunks + i) % kSlots]);
        if (st == PHMM_OK) st = s2;
    }
    h->slot = 0;
    h->defer_d2h = false;
    if (st == PHMM_OK && final_st != PHMM_OK) {
        h->err = "phmm_region_compute: a CIGAR needs more elements than its slot holds (n_out_cigar has the sizes)";
        st = h->err_code = final_st;
    }
    return st;
}

// regions [g0, g1) of validated arguments on one handle (phmm_region_compute_multi: one range per engine): offsets rebased,
// every pointer moved on, nothing gathered
int region_compute_range(phmm_handle *h, const RegionArgs &a, uint32_t g0, uint32_t g1) {
    if (g0 >= g1) return PHMM_OK;
    if (g0 == 0 && g1 == a.n_regions) return region_compute(h, a);
    ChunkView c;
    c.g1 = g0;
    (void)next_chunk(c, g1, a.region_read_off, a.region_hap_off, a.read_off, a.hap_off, a.out_off, true);
    ChunkArgs part;
    part.build(a, c);
    return region_compute(h, part.a);
}

}  // namespace phmm_host

namespace {

RegionArgs pack_args(const phmm_engine_config *cfg, const phmm_realign_config *rcfg, uint32_t n_regions, const uint32_t *region_read_off,
                     const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                     const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *mapq, const uint32_t *read_soft_clip, const uint32_t *hap_off,
                     const uint8_t *hap_bases, const int32_t *region_ref_hap, const uint64_t *out_off, const int32_t *hap_priority,
                     const uint64_t *region_reference_start, const uint32_t *hap_cigar_off, const uint32_t *hap_cigar,
                     const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar, const uint64_t *out_cigar_off,
                     double *out, uint8_t *keep, int32_t *best_allele, double *likelihood, double *confidence, uint32_t *out_cigar,
                     uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status) {
    RegionArgs a;
    a.cfg = *cfg;
    a.rcfg = *rcfg;
    a.n_regions = n_regions;
    a.region_read_off = region_read_off;
    a.region_hap_off = region_hap_off;
    a.read_off = read_off;
    a.read_bases = read_bases;
    a.base_q = base_q;
    a.ins_q = ins_q;
    a.del_q = del_q;
    a.mapq = mapq;
    a.read_soft_clip = read_soft_clip;
    a.hap_off = hap_off;
    a.hap_bases = hap_bases;
    a.region_ref_hap = region_ref_hap;
    a.out_off = out_off;
    a.hap_priority = hap_priority;
    a.region_reference_start = region_reference_start;
    a.hap_cigar_off = hap_cigar_off;
    a.hap_cigar = hap_cigar;
    a.hap_start_wrt_ref = hap_start_wrt_ref;
    a.orig_cigar_off = orig_cigar_off;
    a.orig_cigar = orig_cigar;
    a.out_cigar_off = out_cigar_off;
    a.out = out;
    a.keep = keep;
    a.best_allele = best_allele;
    a.likelihood = likelihood;
    a.confidence = confidence;
    a.out_cigar = out_cigar;
    a.n_out_cigar = n_out_cigar;
    a.new_pos = new_pos;
    a.status = status;
    return a;
}

}  // namespace

namespace phmm_host {
RegionArgs region_pack_args(const phmm_engine_config *cfg, const phmm_realign_config *rcfg, uint32_t n_regions, const uint32_t *region_read_off,
                            const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                            const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *mapq, const uint32_t *read_soft_clip, const uint32_t *hap_off,
                            const uint8_t *hap_bases, const int32_t *region_ref_hap, const uint64_t *out_off, const int32_t *hap_priority,
                            const uint64_t *region_reference_start, const uint32_t *hap_cigar_off, const uint32_t *hap_cigar,
                            const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar, const uint64_t *out_cigar_off,
                            double *out, uint8_t *keep, int32_t *best_allele, double *likelihood, double *confidence, uint32_t *out_cigar,
                            uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status) {
    return pack_args(cfg, rcfg, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, mapq, read_soft_clip, hap_off,
                     hap_bases, region_ref_hap, out_off, hap_priority, region_reference_start, hap_cigar_off, hap_cigar, hap_start_wrt_ref,
                     orig_cigar_off, orig_cigar, out_cigar_off, out, keep, best_allele, likelihood, confidence, out_cigar, n_out_cigar, new_pos, status);
}
}  // namespace phmm_host

extern "C" int phmm_region_compute(phmm_handle *h, const phmm_engine_config *cfg, const phmm_realign_config *rcfg, uint32_t n_regions,
                                   const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                                   const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q,
                                   const uint8_t *mapq, const uint32_t *read_soft_clip, const uint32_t *hap_off, const uint8_t *hap_bases,
                                   const int32_t *region_ref_hap, const uint64_t *out_off, const int32_t *hap_priority,
                                   const uint64_t *region_reference_start, const uint32_t *hap_cigar_off, const uint32_t *hap_cigar,
                                   const uint32_t *hap_start_wrt_ref, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar,
                                   const uint64_t *out_cigar_off, double *out, uint8_t *keep, int32_t *best_allele, double *likelihood,
                                   double *confidence, uint32_t *out_cigar, uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status) {
    if (!h || !cfg || !rcfg) return PHMM_ERR_INVALID_ARG;
    try {
        phmm_host::latch_slot0(h);
        h->err_code = PHMM_OK;
        clear_thread_error(h);
        const RegionArgs a = pack_args(cfg, rcfg, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, mapq,
                                       read_soft_clip, hap_off, hap_bases, region_ref_hap, out_off, hap_priority, region_reference_start,
                                       hap_cigar_off, hap_cigar, hap_start_wrt_ref, orig_cigar_off, orig_cigar, out_cigar_off, out, keep,
                                       best_allele, likelihood, confidence, out_cigar, n_out_cigar, new_pos, status);
        const std::string bad = region_validate(a);
        if (!bad.empty()) return fail(h, bad);
        // The device's resident server takes the call if it is within its limits: staged, one ring entry, a poll (phmm_server.cpp).
        {
            ServerPending *pending = nullptr;
            int st = server_region_submit(h, a, &pending, false);
            if (st == PHMM_OK) {
                st = server_region_wait(h, pending, &h->err, nullptr);
                if (st != kServerRedo) {
                    if (st != PHMM_OK) h->err_code = st;
                    return st;
                }
            } else if (st != kServerNotTaken) {
                return h->err_code = st;
            }
        }
        // (one of many private handles on the device: a one-shot call goes through the device's shared handle -- route_shared)
        const uint32_t nr_all = region_read_off[n_regions];
        if (nr_all && (n_regions < 8 || (size_t)read_off[nr_all] <= one_shot_bytes()))
            if (phmm_handle *via = route_shared(h)) {
                uint64_t ticket = 0;
                int st = phmm_region_submit(via, cfg, rcfg, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, mapq,
                                            read_soft_clip, hap_off, hap_bases, region_ref_hap, out_off, hap_priority, region_reference_start,
                                            hap_cigar_off, hap_cigar, hap_start_wrt_ref, orig_cigar_off, orig_cigar, out_cigar_off, out, keep,
                                            best_allele, likelihood, confidence, out_cigar, n_out_cigar, new_pos, status, &ticket);
                if (st == PHMM_OK) st = phmm_wait(via, ticket);
                if (st != PHMM_OK) {
                    h->err = phmm_last_error(via);
                    h->err_code = st;
                }
                return st;
            }
        return region_compute(h, a);
    } catch (const std::bad_alloc &) {
        h->err = "phmm_region_compute: out of host memory";
        return h->err_code = PHMM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        h->err = std::string("phmm_region_compute: ") + e.what();
        return h->err_code = PHMM_ERR_INTERNAL;
    }
}
