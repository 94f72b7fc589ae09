// C ABI (include/phmm.h) + launch planner of the MI355X PairHMM engine.
//
// Host side of the drop-in boundary: takes the flattened (reads, haplotypes, quals) of any number
// of assembly regions, bins the regions into kernel shape classes <L lanes per pair, K haplotype
// columns per lane>, and launches one gfx950 kernel per class.  Replaces, for the whole batch,
//   PairHMM::initialize            (reference src/pair_hmm/pair_hmm.rs:63-125)   -> phmm_create / plan
//   PairHMM::compute_likelihoods   (:345-375)                                     -> phmm_compute
// There is no CPU fallback anywhere in this file.
#include "../../include/phmm.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <tuple>
#include <unordered_map>
#include <mutex>
#include <string>
#include <system_error>
#include <sched.h>
#include <thread>
#include <vector>

#include "phmm_cigar_internal.hpp"
#include "phmm_host.hpp"
#include <sched.h>
#include "phmm_internal.hpp"
#include "phmm_tables.hpp"

using namespace phmm;

namespace {

#ifndef PHMM_MIXED_RUNS
#define PHMM_MIXED_RUNS 32
#endif
constexpr unsigned kMixedRunsPerSlot = PHMM_MIXED_RUNS;  // runs per wave slot of a mixed batch (see the planner)

std::mutex g_err_mu;
std::string g_create_err = "";
// phmm_submit / phmm_wait are the only entry points several threads may call on one handle, so their messages are kept
// per calling thread; every other entry point resets the pair.
thread_local std::string tl_err;
thread_local const phmm_handle *tl_err_h = nullptr;
thread_local bool tl_internal_create = false;  // phmm_create is making a lane / a backing handle (phmm_host::create_internal)
// the caller's own handles alive per device, and the shared handles their small calls are routed through (phmm_host::route_shared)
std::atomic<int> g_user_handles[kMaxDevices];
std::mutex g_backing_mu;
std::map<std::pair<int, unsigned>, phmm_handle *> g_backing;

constexpr size_t kLdsBytesPerCU = 160 * 1024;
constexpr size_t kLdsRowBytes = 72;  // sizeof(RowConst) in phmm_kernels.hip
constexpr uint32_t kNumSimd = 256 * 4;
constexpr uint64_t kGenericScratchBytes = 1ull << 30;
// Chunked host path: batches whose per-base arrays exceed kOneShotBytes are cut into chunks of regions whose arrays grow
// from kFirstChunkBytes (the GPU starts early) to kChunkBytes (large launches are the efficient ones).  Measured on
// config-2 batches, one shot vs chunked: 32 regions 548 -> 479 us, 256 regions 4.36 -> 3.13 ms, 4096 regions 43.2 ms with
// 4 MB chunks vs 45.5 ms with 512 KB chunks throughout.  (PHMM_CHUNK_KB, PHMM_FIRST_CHUNK_KB, PHMM_ONESHOT_KB: tuning.)
static const size_t kChunkBytes = 4u << 20;
static const size_t kFirstChunkBytes = 512u << 10;
static const size_t kMixedFirstChunkBytes = 4u << 20;  // (swept in round 5, tools/hostpath_ragged_sweep.py: 4 / 8 / 8 / 7 MB was the best schedule)
static const size_t kMixedChunkBytes = 8u << 20;
static const size_t kOneShotBytes = 512u << 10;
static const size_t kStageInBytes = 640u << 10;
static const size_t kZeroCopyOutBytes = 160u << 10;
static const int kForcedEagerD2H = getenv("PHMM_EAGER_D2H") ? atoi(getenv("PHMM_EAGER_D2H")) : -1;
// (these six are tuning knobs of the host path, latched when the library is loaded; everything else is in Switches)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Entry points leave the calling thread's current HIP device as they found it.
struct DeviceGuard {
    int prev = -1, dev;
    bool ok = true;
    explicit DeviceGuard(int device) : dev(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

struct ShapeClass {
    int L = 0, K = 0;  // L == 0 -> generic kernel
    std::vector<uint32_t> reads;  // global read indices (host copy; uploaded unless identity)
    bool identity = false;        // reads == 0..n-1
    uint32_t max_r = 0, max_h = 0, max_quads = 0;
    uint64_t cells = 0;
    // launch configuration
    uint32_t lds_rows = 8;
    int waves_per_block = MAX_WAVES_PER_BLOCK;
    size_t lds_bytes = 0;
    dim3 grid;
    // chained class: items are (region, haplotype group, run of reads) instead of single reads
    bool chain = false;
    std::vector<uint32_t> regions;  // member regions (chain classes)
    std::vector<ChainItem> chain_items;  // host side; launched as part of its ChainGroup
    uint32_t cnd_select = 0;
    int streams = 1;  // chained classes: sub-runs swept side by side (phmm_chain_kernels.hip)
    bool f32_first = false;  // chained class at 16 lanes per pair of a PHMM_FLAG_F32_FIRST handle: f32 sweep, then the
                             // f64 per-read kernel over the reads it flagged (the per-read launch geometry is filled in too)
    // device
    uint32_t *d_reads = nullptr;
    // generic only
    std::vector<uint64_t> pair_first;
    uint64_t *d_pair_first = nullptr;
    double *d_scratch = nullptr;
    uint32_t generic_blocks = 0;
    char name[48] = {0};
};

}  // namespace

struct phmm_batch {
    phmm_handle *h = nullptr;
    uint32_t n_regions = 0, n_reads = 0, n_haps = 0;
    uint64_t n_out = 0, read_bytes = 0, hap_bytes = 0;
    uint64_t cells = 0, alg_bytes = 0;
    std::vector<ShapeClass> classes;
    // every chained f64 class of one lanes-per-pair value goes out in ONE launch (phmm_chain_kernels.hip)
    struct ChainGroup {
        int L = 0;
        bool f32 = false;  // the f32 sweep of a PHMM_FLAG_F32_FIRST handle (the f64 per-read redo follows per class)
        int single_k = 0;  // the K all items share (per-K kernel), 0 = mixed (any-K kernel)
        std::vector<ChainItem> items;
        ChainItem *d_items = nullptr;
    };
    std::vector<ChainGroup> chain_groups;
    // device metadata (one allocation)
    uint32_t *d_read_region = nullptr, *d_region_read_off = nullptr, *d_region_hap_off = nullptr, *d_read_off = nullptr,
             *d_hap_off = nullptr, *d_status = nullptr;
    uint64_t *d_out_off = nullptr;
    uint8_t *d_redo = nullptr;  // [n_reads] f32-first mode: reads the f64 per-read kernel has to redo
    // payload
    const uint8_t *d_read_bases = nullptr, *d_base_q = nullptr, *d_ins_q = nullptr, *d_del_q = nullptr, *d_gcp = nullptr,
                  *d_hap_bases = nullptr;
    double *d_out = nullptr;
    void *d_owned = nullptr;  // batch-owned payload + out (phmm_batch_upload)
    Arena *arena = nullptr;   // non-null: every device allocation of this batch lives in the handle's arena
    hipStream_t home_stream = nullptr;  // the handle stream this batch was staged on
    std::vector<void *> mallocs;  // hipMalloc'ed pieces owned by this batch (persistent batches, generic scratch)
    size_t out_arena_off = 0;     // arena mode: offset of [status word | out]
    bool tight_out = true;        // out_off has no gaps (every slot is written by a kernel)
    std::vector<std::pair<uint64_t, uint64_t>> out_extents;  // !tight_out: (first slot, Nr*Nh) per region -- gaps stay untouched
    uint32_t max_h = 0;           // longest haplotype (sizes the scratch of phmm_rescue)
    double *rescue_scratch = nullptr;  // non-null: the exact pass rides behind every launch (persistent batches: own
                                       // scratch; engine-level call: the arena's, its results are consumed on the device)
    uint32_t rescue_blocks = 0;
    bool bound = false;
    std::string dominant;
    uint64_t pad_column_cells = 0, pad_slot_cells = 0;  // ... of which columns beyond a haplotype's end / haplotype slots left empty
    uint64_t swept_cells = 0;  // lane-cells the planned launches sweep: every row of every wave x 64 lanes x its K columns, padding
                               // columns, empty haplotype slots and all (phmm_batch_executed_cells)
};

namespace {

bool hip_ok(phmm_handle *h, hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (h) {
        h->err = buf;
        h->err_code = PHMM_ERR_HIP;
    }
    else {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_create_err = buf;
    }
    return false;
}
#define HIP_TRY(h, call, ret)                  \
    do {                                       \
        if (!hip_ok((h), (call), #call)) return ret; \
    } while (0)

// No C++ exception crosses the C ABI: every extern "C" body that can allocate runs inside PHMM_GUARD.
int on_exception(phmm_handle *h, const char *where, const char *what, int code) {
    std::string msg = std::string(where) + ": " + what;
    if (h) {
        h->err = msg;
        h->err_code = code;
    } else {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_create_err = msg;
    }
    return code;
}
#define PHMM_GUARD_BEGIN try {
#define PHMM_GUARD_END(h, where, fail)                                                                    \
    }                                                                                                     \
    catch (const std::bad_alloc &) {                                                                      \
        (void)on_exception((h), (where), "out of host memory", PHMM_ERR_NO_MEMORY);                       \
        return fail(PHMM_ERR_NO_MEMORY);                                                                  \
    }                                                                                                     \
    catch (const std::exception &e) {                                                                     \
        (void)on_exception((h), (where), e.what(), PHMM_ERR_INTERNAL);                                    \
        return fail(PHMM_ERR_INTERNAL);                                                                   \
    }                                                                                                     \
    catch (...) {                                                                                         \
        (void)on_exception((h), (where), "unknown exception", PHMM_ERR_INTERNAL);                         \
        return fail(PHMM_ERR_INTERNAL);                                                                   \
    }
#define PHMM_FAIL_CODE(c) (c)
#define PHMM_FAIL_NULL(c) nullptr

int round_up_k(int k) {
    for (int i = 0; i < kNumInstantiatedK; ++i)
        if (kInstantiatedK[i] >= k) return kInstantiatedK[i];
    return 0;
}

// Registers cap the resident waves per SIMD (3*K f64 of DP state per lane dominates; K <= 25 is
// compiled for 2 waves, PHMM_TWO_WAVE_MAX_K).
int waves_per_simd(int K) { return K <= 25 ? 2 : 1; }

// Throughput model of a region under <L,K>, calibrated on MI355X (tools/shapes.py): useful fraction of
// issued lane-steps x the per-step overhead (DPP shifts, LDS fetch, loop: ~11 of 7*K+11 VALU ops per
// step) x the issue rate one resident wave reaches alone (a wave issues a VALU op every ~6 clk, two waves
// together one every ~4.7: tools/ubench/issue.hip; measured 0.81 on <16,25>).
// Chained kernel at 16 lanes per pair: the four haplotype slots of a wave can be shared by S = 1, 2 or 4 streams of
// reads (phmm_chain_kernels.hip), so any haplotype count fills them.  Every extra stream costs row-producer work
// (rows are built per stream, in shorter ticks): measured 3990 / 3700 / 3300 GCUPS at 1 / 2 / 4 streams with all
// slots busy, i.e. ~6 % per extra stream.  Returns S, and the slot fill (times that factor) it achieves.
int chain_streams(uint32_t nh, double *fill_out) {
    int best_s = 1;
    double best = 0.0;
    for (int S : {1, 2, 4}) {
        const uint32_t gs = 4 / S;
        const double fill = (double)nh / (double)(((nh + gs - 1) / gs) * gs) * (1.0 - 0.06 * (S - 1));
        if (fill > best + 1e-9) {
            best = fill;
            best_s = S;
        }
    }
    if (fill_out) *fill_out = best;
    return best_s;
}

double shape_efficiency(int L, int K, uint32_t nh, uint32_t mean_r, uint32_t max_h, bool chained) {
    const int G = WAVE / L;
    double hap_fill = (double)nh / (double)(((nh + G - 1) / G) * G);
    if (chained && L == 16) (void)chain_streams(nh, &hap_fill);
    // fill / drain steps of the lane pipeline: per read, or (chained kernel) amortised over a run of reads
    const double ramp = chained ? 1.0 : (double)std::max<uint32_t>(mean_r, 1) / (double)(std::max<uint32_t>(mean_r, 1) + L - 1);
    const double col_fill = (double)max_h / (double)(L * K);
    const double step = 7.0 * K / (7.0 * K + 11.0);
    const double occ = waves_per_simd(K) >= 2 ? 1.0 : 0.84;
    return hap_fill * ramp * col_fill * step * occ;
}

}  // namespace

// The one place the PHMM_* developer switches are read (phmm_set_switch changes them per handle afterwards): phmm_create,
// and phmm_plan_describe for its host-only handle, so that a described plan is the plan an engine of this process makes.
static void read_env_switches(Switches &w) {
    auto env = [](const char *name, int &dst) {
        if (const char *e = getenv(name)) dst = atoi(e);
    };
    env("PHMM_FORCE_L", w.force_L);
    env("PHMM_FORCE_CHAIN", w.force_chain);
    env("PHMM_FORCE_STREAMS", w.force_streams);
    env("PHMM_SW_LITE", w.sw_lite);
    env("PHMM_SW_CHUNKS", w.sw_chunks);
    env("PHMM_SW_LANES", w.sw_lanes);
    env("PHMM_SW_TRANSPOSE", w.sw_transpose);
    env("PHMM_REGION_SW_ALL", w.region_sw_all);
    env("PHMM_REGION_SERVER", w.region_server);
    env("PHMM_SERVER_IDLE_US", w.server_idle_us);
    env("PHMM_SERVER_TRACE", w.server_trace);
    env("PHMM_REGION_FLAG_WAIT", w.region_flag_wait);
    env("PHMM_REGION_OWN_QUEUE", w.region_own_queue);
    env("PHMM_REGION_PICK_TIMEOUT_US", w.region_pick_timeout_us);
    env("PHMM_REGION_DEBUG_PICK", w.region_debug_pick);
    env("PHMM_MIRROR_CANARY", w.mirror_canary);
    env("PHMM_ROUTE_SHARED", w.route_shared);
    w.sw_no_zero_copy = getenv("PHMM_SW_NO_ZERO_COPY") != nullptr;
    w.sw_clock = getenv("PHMM_SW_CLOCK") != nullptr;
    w.no_pipeline = getenv("PHMM_NO_PIPELINE") != nullptr;
    w.no_rescue = getenv("PHMM_NO_RESCUE") != nullptr;
    w.trace = getenv("PHMM_TRACE") != nullptr;
}

extern "C" {

int phmm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *phmm_last_error(phmm_handle *h) {
    if (h && tl_err_h == h) return tl_err.c_str();
    if (h) return h->err.c_str();
    std::lock_guard<std::mutex> g(g_err_mu);
    return g_create_err.c_str();
}

phmm_handle *phmm_create(int device_id, unsigned flags) {
    int n = phmm_device_count();
    if (device_id < 0 || device_id >= n || device_id >= kMaxDevices) {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_create_err = "phmm_create: no HIP device with that id (this engine has no CPU fallback)";
        return nullptr;
    }
    DeviceGuard dg(device_id);
    if (!dg.ok && !hip_ok(nullptr, hipErrorInvalidDevice, "hipSetDevice")) return nullptr;
    PHMM_GUARD_BEGIN
    phmm_handle *h = new phmm_handle();
    h->device = device_id;
    h->flags = flags;
    h->internal = tl_internal_create;
    if (!h->internal) g_user_handles[device_id].fetch_add(1, std::memory_order_relaxed);
    phmm_host::handle_born(h);
    read_env_switches(h->sw);
    const auto &eps = table_eps();
    const auto &eps3 = table_eps_third();
    const auto &mm = table_match_to_match();
    bool ok = true;
    // (slot 0 -- the stream of every call that is one enqueue -- is a hardware queue of the handle's own where the device's pool
    // has one left: phmm_region.cpp, queues_acquire)
    for (int i = 0; i < kSlots && ok; ++i)
        ok = hip_ok(nullptr, hipStreamCreateWithFlags(&h->streams[i], hipStreamNonBlocking), "hipStreamCreate");
    h->stream0_ordinary = h->streams[0];
    ok = ok &&
              hip_ok(nullptr, hipMalloc(&h->d_eps, 256 * sizeof(double)), "hipMalloc eps") &&
              hip_ok(nullptr, hipMalloc(&h->d_eps_mis, 256 * sizeof(double)), "hipMalloc eps_mis") &&
              hip_ok(nullptr, hipMalloc(&h->d_mm, mm.size() * sizeof(double)), "hipMalloc mm") &&
              hip_ok(nullptr, hipMemcpy(h->d_eps, eps.data(), 256 * sizeof(double), hipMemcpyHostToDevice), "copy eps") &&
              hip_ok(nullptr,
                     hipMemcpy(h->d_eps_mis, (flags & PHMM_FLAG_NO_TRISTATE) ? eps.data() : eps3.data(),
                               256 * sizeof(double), hipMemcpyHostToDevice),
                     "copy eps_mis") &&
              hip_ok(nullptr, hipMemcpy(h->d_mm, mm.data(), mm.size() * sizeof(double), hipMemcpyHostToDevice), "copy mm");
    if (ok) {  // derived tables of the pre-scaled row form (phmm_device.hpp, RowConst)
        std::vector<double> ratio(256, 0.0), inv_om(256, 0.0);
        const auto &mis = (flags & PHMM_FLAG_NO_TRISTATE) ? eps : eps3;
        for (int q = 1; q < 256; ++q) {
            ratio[q] = mis[q] / (1.0 - eps[q]);
            inv_om[q] = 1.0 / (1.0 - eps[q]);
        }
        ok = hip_ok(nullptr, hipMalloc(&h->d_ratio_mis, 256 * sizeof(double)), "hipMalloc ratio_mis") &&
             hip_ok(nullptr, hipMalloc(&h->d_inv_om, 256 * sizeof(double)), "hipMalloc inv_om") &&
             hip_ok(nullptr, hipMemcpy(h->d_ratio_mis, ratio.data(), 256 * sizeof(double), hipMemcpyHostToDevice), "copy ratio_mis") &&
             hip_ok(nullptr, hipMemcpy(h->d_inv_om, inv_om.data(), 256 * sizeof(double), hipMemcpyHostToDevice), "copy inv_om");
    }
    if (ok) {
        std::vector<unsigned char> all(4 * 128, 0);
        for (int m = 1; m <= 3; ++m) {
            const auto c = pcr_error_model_cache(m);
            std::copy(c.begin(), c.end(), all.begin() + m * 128);
        }
        ok = hip_ok(nullptr, hipMalloc((void **)&h->d_pcr_cache, all.size()), "hipMalloc pcr") &&
             hip_ok(nullptr, hipMemcpy(h->d_pcr_cache, all.data(), all.size(), hipMemcpyHostToDevice), "copy pcr");
    }
    if (!ok) {
        phmm_destroy(h);
        return nullptr;
    }
    return h;
    PHMM_GUARD_END(nullptr, "phmm_create", PHMM_FAIL_NULL)
}

void phmm_destroy(phmm_handle *h) {
    if (!h) return;
    if (h->comb) phmm_host::combiner_destroy(h->comb);
    DeviceGuard dg(h->device);
    h->streams[0] = h->stream0_ordinary;
    phmm_host::queues_release(h);
    phmm_host::handle_died(h);
    for (int i = 0; i < kSlots; ++i)
        if (h->streams[i]) (void)hipStreamDestroy(h->streams[i]);
    if (h->d_eps) (void)hipFree(h->d_eps);
    if (h->d_eps_mis) (void)hipFree(h->d_eps_mis);
    if (h->d_mm) (void)hipFree(h->d_mm);
    if (h->d_ratio_mis) (void)hipFree(h->d_ratio_mis);
    if (h->d_inv_om) (void)hipFree(h->d_inv_om);
    if (h->d_pcr_cache) (void)hipFree(h->d_pcr_cache);
    for (int i = 0; i < kSlots; ++i) {
        if (h->arenas[i].dev) (void)hipFree(h->arenas[i].dev);
        if (h->arenas[i].host) (void)hipHostFree(h->arenas[i].host);
        if (h->arenas[i].rescue) (void)hipFree(h->arenas[i].rescue);
    }
    if (h->swork.dev) (void)hipFree(h->swork.dev);
    if (h->swork.host) (void)hipHostFree(h->swork.host);
    if (h->swork.slab) (void)hipFree(h->swork.slab);
    if (h->swork.ws) (void)hipFree(h->swork.ws);
    if (h->swork.ext) (void)hipFree(h->swork.ext);
    for (int c = 0; c < phmm_handle::SwWork::kMaxChunks; ++c)
        for (hipEvent_t e : {h->swork.ev_in[c], h->swork.ev_out[c], h->swork.ev_k0[c], h->swork.ev_k1[c]})
            if (e) (void)hipEventDestroy(e);
    if (h->swork.region_sw_done) (void)hipEventDestroy(h->swork.region_sw_done);
    if (h->swork.ev_second) (void)hipEventDestroy(h->swork.ev_second);
    if (h->swork.d_pair_done) (void)hipFree(h->swork.d_pair_done);
    phmm_host::halves_release(h);
    for (int i = 0; i < 2; ++i) {
        if (h->swork.all_stream[i]) (void)hipStreamDestroy(h->swork.all_stream[i]);
        if (h->swork.pair_main[i]) (void)hipStreamDestroy(h->swork.pair_main[i]);
    }
    for (int i = 0; i < phmm_handle::kSideStreams; ++i) {
        if (h->side_streams[i]) (void)hipStreamDestroy(h->side_streams[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    const bool internal = h->internal;
    const int device = h->device;
    delete h;
    if (internal) return;
    // The count reaching zero and the backing handles leaving the map are ONE step under the map's lock (route_shared looks
    // them up under the same lock): a handle created after this step finds no entry and makes its own backing handle; one
    // created before it keeps the count above zero.
    std::vector<phmm_handle *> gone;
    bool last_of_the_callers = false;
    {
        std::lock_guard<std::mutex> lk(g_backing_mu);
        last_of_the_callers = g_user_handles[device].fetch_sub(1, std::memory_order_acq_rel) == 1;
        if (last_of_the_callers) {
            for (auto it = g_backing.begin(); it != g_backing.end();) {
                if (it->first.first == device) {
                    gone.push_back(it->second);
                    it = g_backing.erase(it);
                } else {
                    ++it;
                }
            }
        }
    }
    if (last_of_the_callers) {  // nobody is left to route: the device's backing handles go too
        for (phmm_handle *b : gone) phmm_destroy(b);
        phmm_host::server_quiesce(device);  // (the region server leaves the chip by itself once idle: wait for that)
    }
}

size_t phmm_table_eps(const double **eps) {
    *eps = table_eps().data();
    return table_eps().size();
}
size_t phmm_table_match_to_match(const double **mm) {
    *mm = table_match_to_match().data();
    return table_match_to_match().size();
}

void phmm_batch_destroy(phmm_batch *b) {
    if (!b) return;
    DeviceGuard dg(b->h->device);
    for (void *m : b->mallocs) (void)hipFree(m);
    delete b;
}

}  // extern "C"

namespace phmm_host {

void set_thread_error(const phmm_handle *h, const std::string &msg) {
    tl_err = msg;
    tl_err_h = h;
}
void clear_thread_error(const phmm_handle *h) {
    if (tl_err_h == h) tl_err_h = nullptr;
}

phmm_handle *create_internal(int device, unsigned flags) {
    struct Mark {
        bool prev = tl_internal_create;
        Mark() { tl_internal_create = true; }
        ~Mark() { tl_internal_create = prev; }
    } mark;
    return phmm_create(device, flags);
}

int user_handles_on(int device) { return g_user_handles[device % kMaxDevices].load(std::memory_order_relaxed); }

// The wait of a one-shot call.  hipStreamSynchronize spins; with more caller threads than the process has cores -- its affinity
// mask or its container's CPU quota: 16 of 256 on the GPU box -- the spinning waiters are throttled together with the callers
// that have something to stage, and 32 private handles ran at HALF the rate of 16 (phmm_compute 46 -> 22 k regions/s, VERDICT r4;
// round 5 hid it by routing such handles through the shared combiner; the region server's waiters showed the same: 22.9 k
// spinning, 45.6 k with short sleeps, NOTEBOOK 20.2).  So: when the caller holds more handles on the device than it has cores,
// look at the stream, sleep 20 us, look again: 24 / 32 handles 31 / 22 -> 46 / 45 k, nothing routed, nothing combined.
int process_cores() {
    static const int cores = [] {
        cpu_set_t set;
        CPU_ZERO(&set);
        int n = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // (a container's CPU quota: spinning beyond it is throttled)
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) n = std::min<int>(n, (int)((quota + period - 1) / period));
            fclose(f);
        }
        return n > 0 ? n : 1;
    }();
    return cores;
}
bool more_callers_than_cores(const phmm_handle *h) { return !h->internal && user_handles_on(h->device) > process_cores(); }
hipError_t wait_stream(const phmm_handle *h, hipStream_t s) {
    if (!more_callers_than_cores(h)) return hipStreamSynchronize(s);
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

phmm_handle *route_shared(phmm_handle *h) {
    // (opt-in since round 6: which regions share a flush depends on timing, so routed results are reproducible to ~1e-13, not bit
    // for bit -- the default for many private handles is the region server, whose results never depend on the load)
    if (h->internal || h->sw_touched || h->comb || h->sw.route_shared <= 0) return nullptr;
    const int above = h->sw.route_shared;
    if (g_user_handles[h->device].load(std::memory_order_relaxed) <= above) return nullptr;
    if (h->backing) return h->backing;  // (lives as long as one of the caller's handles does on the device: at least as long as h)
    std::lock_guard<std::mutex> lk(g_backing_mu);
    phmm_handle *&b = g_backing[{h->device, h->flags}];
    if (!b) {
        b = create_internal(h->device, h->flags);
        if (!b) {
            g_backing.erase({h->device, h->flags});
            return nullptr;  // (no memory for another engine: the call stays where it is)
        }
    }
    return h->backing = b;
}

const char *validate_offsets(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                             const uint32_t *read_off, const uint32_t *hap_off, const uint64_t *out_off, bool *tight_out) {
    if (!region_read_off || !region_hap_off || !read_off || !hap_off || !out_off) return "phmm_batch_create: null offset array";
    if (region_read_off[0] != 0 || region_hap_off[0] != 0 || read_off[0] != 0 || hap_off[0] != 0 || out_off[0] != 0)
        return "phmm_batch_create: offset arrays must start at 0";
    const uint32_t n_reads = region_read_off[n_regions], n_haps = region_hap_off[n_regions];
    bool tight = true;
    for (uint32_t g = 0; g < n_regions; ++g) {
        if (region_read_off[g + 1] < region_read_off[g] || region_hap_off[g + 1] < region_hap_off[g])
            return "phmm_batch_create: region offsets not monotonic";
        const uint64_t need = (uint64_t)(region_read_off[g + 1] - region_read_off[g]) *
                              (uint64_t)(region_hap_off[g + 1] - region_hap_off[g]);
        if (out_off[g + 1] < out_off[g] || out_off[g + 1] - out_off[g] < need)
            return "phmm_batch_create: out_off leaves too little room for a region (needs Nr*Nh doubles)";
        if (out_off[g + 1] - out_off[g] != need) tight = false;
    }
    for (uint32_t r = 0; r < n_reads; ++r)
        if (read_off[r + 1] < read_off[r]) return "phmm_batch_create: read_off not monotonic";
    for (uint32_t a = 0; a < n_haps; ++a) {
        if (hap_off[a + 1] < hap_off[a]) return "phmm_batch_create: hap_off not monotonic";
        // the reference would compute 2^1020 / 0 (pair_hmm.rs:515-517); SURVEY 8b asks for an explicit error
        if (hap_off[a + 1] == hap_off[a]) return "phmm_batch_create: empty haplotype";
    }
    if (tight_out) *tight_out = tight;
    return nullptr;
}

}  // namespace phmm_host
using namespace phmm_host;

// (`dry`: plan only -- no device is touched, every "device" pointer of the batch stays null: phmm_plan_describe)
static phmm_batch *batch_create_impl(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off,
                                     const uint32_t *region_hap_off, const uint32_t *read_off,
                                     const uint32_t *hap_off, const uint64_t *out_off, bool use_arena,
                                     size_t extra_arena_bytes = 0, bool dry = false) {
    if (!h) return nullptr;
    h->err.clear();
    h->err_code = PHMM_OK;
    if (tl_err_h == h) tl_err_h = nullptr;
    bool tight = true;
    if (const char *bad = validate_offsets(n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, &tight)) {
        h->err = bad;
        h->err_code = PHMM_ERR_INVALID_ARG;
        return nullptr;
    }
    const uint32_t n_reads = region_read_off[n_regions], n_haps = region_hap_off[n_regions];
    std::unique_ptr<DeviceGuard> dg;
    if (!dry) {
        dg.reset(new DeviceGuard(h->device));
        if (!dg->ok) {
            h->err = "hipSetDevice failed";
            h->err_code = PHMM_ERR_HIP;
            return nullptr;
        }
    }

    // (owned until the plan is complete: an exception on the way -- a host allocation -- releases it and what it holds)
    struct BatchDeleter {
        bool dry;
        void operator()(phmm_batch *x) const {
            if (!dry)
                for (void *m : x->mallocs) (void)hipFree(m);
            delete x;
        }
    };
    std::unique_ptr<phmm_batch, BatchDeleter> owner(new phmm_batch(), BatchDeleter{dry});
    phmm_batch *b = owner.get();
    b->h = h;
    b->n_regions = n_regions;
    b->n_reads = n_reads;
    b->n_haps = n_haps;
    b->n_out = out_off[n_regions];
    b->read_bytes = read_off[n_reads];
    b->hap_bytes = hap_off[n_haps];
    b->tight_out = tight;
    if (!tight)
        for (uint32_t g = 0; g < n_regions; ++g)
            b->out_extents.emplace_back(out_off[g], (uint64_t)(region_read_off[g + 1] - region_read_off[g]) *
                                                        (uint64_t)(region_hap_off[g + 1] - region_hap_off[g]));
    b->home_stream = h->S();

    bool ok = true;
    auto mark_now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double marks[8] = {mark_now()};
    int n_marks = 1;
    auto mark = [&]() { if (n_marks < 8) marks[n_marks++] = mark_now(); };

    // ---- per-region shape, totals -----------------------------------------------------------
    struct RegionShape {
        uint32_t nr, nh, max_r, max_h, mean_r, min_r = 0xffffffffu, min_h = 0xffffffffu;
        uint64_t cells;
    };
    std::vector<RegionShape> shape(n_regions);
    std::vector<uint32_t> read_region(n_reads);
    for (uint32_t g = 0; g < n_regions; ++g) {
        RegionShape s{};
        s.nr = region_read_off[g + 1] - region_read_off[g];
        s.nh = region_hap_off[g + 1] - region_hap_off[g];
        uint64_t sum_r = 0, sum_h = 0;
        for (uint32_t r = region_read_off[g]; r < region_read_off[g + 1]; ++r) {
            const uint32_t len = read_off[r + 1] - read_off[r];
            s.max_r = std::max(s.max_r, len);
            s.min_r = std::min(s.min_r, len);
            sum_r += len;
            read_region[r] = g;
        }
        for (uint32_t a = region_hap_off[g]; a < region_hap_off[g + 1]; ++a) {
            const uint32_t len = hap_off[a + 1] - hap_off[a];
            s.max_h = std::max(s.max_h, len);
            s.min_h = std::min(s.min_h, len);
            sum_h += len;
        }
        s.mean_r = s.nr ? (uint32_t)(sum_r / s.nr) : 0;
        s.cells = sum_r * sum_h;
        b->cells += s.cells;
        b->alg_bytes += 5 * sum_r + sum_h + 8ull * s.nr * s.nh;
        shape[g] = s;
    }

    mark();  // 1: shapes
    // ---- choose <L,K> per region ------------------------------------------------------------
    // Candidates L in {16,32,64}; K = ceil(max_h / L) rounded up to an instantiated value.
    // Pick the most efficient one, then trade lanes-per-pair for more waves while the batch is
    // too small to fill the chip.
    // The chained kernel holds a 19 KB LDS ring per wave (two waves per SIMD): a win wherever the per-read kernel
    // runs two waves per SIMD anyway, a loss against the three or four waves small K gets at 32 / 64 lanes per
    // pair (measured: <32,10> 3150 per-read vs 2860 chained; <32,13> 3030 vs 3450; <32,19> 3220 vs 3470).
    const Switches &sw = h->sw;
    const bool chain_forced = sw.force_chain >= 0;  // tests: every chainable shape chains
    auto chain_shape_ok = [&](int L, int k, const RegionShape &s) {
        return k > 0 && k <= chain_max_k() && (L == 16 || k >= 13 || chain_forced) && s.min_r >= 1 && s.min_h >= 1 &&
               s.nh <= 0xffffu /* ChainItem::quad */;
    };
    bool assume_chain = false;  // second planning pass: the batch is large enough for the chained kernel
    auto pick = [&](const RegionShape &s, int min_L, int &L_out, int &K_out) {
        double best = -1.0;
        L_out = 0;
        K_out = 0;
        for (int L : {16, 32, 64}) {
            if (L < min_L) continue;
            if (h->sw.force_L && L != h->sw.force_L) continue;
            const int k = round_up_k((int)((std::max<uint32_t>(s.max_h, 1) + L - 1) / L));
            if (!k) continue;
            const double e = shape_efficiency(L, k, s.nh, s.mean_r, s.max_h, assume_chain && chain_shape_ok(L, k, s));
            if (e > best) {
                best = e;
                L_out = L;
                K_out = k;
            }
        }
    };
    std::vector<int> reg_L(n_regions), reg_K(n_regions);
    int min_L = 16;
    auto plan_shapes = [&]() {
        min_L = 16;
        for (;;) {
            uint64_t waves = 0;
            for (uint32_t g = 0; g < n_regions; ++g) {
                const RegionShape &s = shape[g];
                if (!s.nr || !s.nh) {
                    reg_L[g] = reg_K[g] = -1;  // nothing to do
                    continue;
                }
                pick(s, min_L, reg_L[g], reg_K[g]);
                if (reg_L[g]) waves += (uint64_t)s.nr * ((s.nh + WAVE / reg_L[g] - 1) / (WAVE / reg_L[g]));
            }
            // one wave per SIMD is enough to stop trading lanes for waves (measured on 1, 2, 4 regions of config 2:
            // <64,5> 41 us, <32,10> 54 us vs <64,5> 60 us, <16,19> 87 us vs <32,10> 88 us)
            if (waves * h->gpu_sharers >= 1ull * kNumSimd || min_L == 64 || h->sw.force_L) break;
            min_L *= 2;
        }
    };
    plan_shapes();

    // Chained kernel (phmm_chain_kernels.hip): reads of a region stream back to back through the lane
    // pipeline, which removes the per-read fill/drain steps.  Worth it (and balanced) only when there is
    // enough work to give every wave a run of reads: decide per batch, qualify per region.
    const int force_streams = sw.force_streams;  // tests: 1 | 2 | 4
    auto streams_of = [&](uint32_t g) {
        if (reg_L[g] != 16) return 1;
        if (force_streams == 1 || force_streams == 2 || force_streams == 4) return force_streams;
        return chain_streams(shape[g].nh, nullptr);
    };
    auto count_units = [&]() {  // wave-sweeps (one read against one wave-load of haplotypes) under the chosen shapes
        uint64_t u = 0;
        for (uint32_t g = 0; g < n_regions; ++g)
            if (reg_L[g] > 0) {
                const uint32_t S = (uint32_t)streams_of(g), gs = (uint32_t)(WAVE / reg_L[g]) / S;
                u += (uint64_t)shape[g].nr * ((shape[g].nh + gs - 1) / gs) / S;
            }
        return u;
    };
    uint64_t units = count_units();
    // run length: about eight runs per wave slot (balance), but never runs shorter than four reads (measured on 128
    // regions of config 2: runs of 2 reads 3380, of 4 reads 3530, per-read kernel 3450 GCUPS); below two runs of two
    // per slot the batch stays with the per-read kernel
    auto runs_for = [&](uint64_t u) {
        const uint32_t r = (uint32_t)std::min<uint64_t>(CHAIN_MAX_READS, u / (8ull * 2 * kNumSimd));
        return r >= 2 && r < 4 ? 4u : r;
    };
    uint32_t chain_reads = runs_for(units);
    if (chain_forced) chain_reads = (uint32_t)std::min(CHAIN_MAX_READS, sw.force_chain);
    if (chain_reads >= 2 && !h->sw.force_L) {
        // chained sweeps pay no per-read fill/drain: choose the shapes again without that term (more lanes per pair
        // become attractive for regions with few haplotypes), and keep the result if the batch still chains
        std::vector<int> L0 = reg_L, K0 = reg_K;
        assume_chain = true;
        plan_shapes();
        const uint32_t cr = chain_forced ? chain_reads : runs_for(count_units());
        if (cr >= 2 && min_L == 16) {
            chain_reads = cr;
            units = count_units();
        } else {
            reg_L = L0;
            reg_K = K0;
        }
        assume_chain = false;
    }
    auto chainable = [&](uint32_t g) {
        const RegionShape &s = shape[g];
        return chain_reads >= 2 && reg_L[g] > 0 && chain_shape_ok(reg_L[g], reg_K[g], s);
    };

    if (sw.trace)
        fprintf(stderr, "phmm plan: %u regions, min_L %d, units %llu, chain_reads %u, region0 <%d,%d> chainable %d\n", n_regions,
                min_L, (unsigned long long)units, chain_reads, n_regions ? reg_L[0] : 0, n_regions ? reg_K[0] : 0,
                n_regions ? (int)chainable(0) : 0);
    mark();  // 2: <L,K> choice
    std::map<std::tuple<int, int, int>, ShapeClass> by_shape;  // (L, K, 0 = per-read kernel | streams of the chained kernel)
    for (uint32_t g = 0; g < n_regions; ++g) {
        if (reg_L[g] < 0) continue;
        const RegionShape &s = shape[g];
        int L = reg_L[g], K = reg_K[g];
        // LDS staging must hold the longest read of the region, one wave per block at least
        const size_t rows = align_up((size_t)s.max_r + 1, 8);
        if (L && rows * kLdsRowBytes > kLdsBytesPerCU) L = K = 0;
        const bool chain = L && chainable(g);
        const int streams = chain ? streams_of(g) : 1;
        ShapeClass &c = by_shape[std::make_tuple(L, K, chain ? streams : 0)];
        c.L = L;
        c.K = K;
        c.chain = chain;
        c.streams = streams;
        if (chain) c.regions.push_back(g);
        for (uint32_t r = region_read_off[g]; r < region_read_off[g + 1]; ++r) c.reads.push_back(r);
        c.max_r = std::max(c.max_r, s.max_r);
        c.max_h = std::max(c.max_h, s.max_h);
        if (L) c.max_quads = std::max(c.max_quads, (s.nh + WAVE / L - 1) / (WAVE / L));
        c.cells += s.cells;
        if (!L)
            for (uint32_t r = region_read_off[g]; r < region_read_off[g + 1]; ++r) c.pair_first.push_back(s.nh);
    }

    // Reads per run.  Uniform batches get `chain_reads` (about eight runs per wave slot), mixed ones a quarter of that (below).
    // Scaling a region's count by its cost per read -- (rows + SUM + RESET) x (7 VALU per column + ~11 per step) relative to
    // the batch's mean, so that every work item costs about the same -- looked right and measured wrong once the items were
    // sorted by cost and spread over the XCDs (1 536 mixed regions: 17.4 ms with it, 16.8 without): short runs of expensive
    // reads pay the pipeline's fill more often than they save at the tail.  PHMM_COST_SCALED_RUNS builds it back in (A/B).
    std::vector<uint32_t> reg_run(n_regions, 0);
    {
#ifdef PHMM_COST_SCALED_RUNS
        auto read_cost = [&](uint32_t g, int K) { return (double)(shape[g].mean_r + 2) * (7.0 * K + 11.0); };
        double cost_sum = 0.0, unit_sum = 0.0;
        for (const auto &kv : by_shape)
            if (kv.second.chain)
                for (uint32_t g : kv.second.regions) {
                    const uint32_t gs = (uint32_t)(WAVE / kv.second.L) / (uint32_t)kv.second.streams;
                    const double u = (double)shape[g].nr * ((shape[g].nh + gs - 1) / gs) / kv.second.streams;
                    cost_sum += u * read_cost(g, kv.second.K);
                    unit_sum += u;
                }
        const double mean_cost = unit_sum > 0 ? cost_sum / unit_sum : 1.0;
#endif
        // A uniform batch balances with eight equal runs per wave slot; a mix of classes does not -- its items differ in
        // cost whatever the estimate, and the launch ends when the last long item does.  Mixed batches therefore get runs
        // a quarter as long (32 per slot, never below 4 reads): 1 536 mixed regions 20.4 -> 16.9 ms.
        size_t n_chain_classes = 0;
        for (const auto &kv : by_shape) n_chain_classes += kv.second.chain ? 1 : 0;
        uint32_t base_reads = chain_reads;
        if (n_chain_classes > 1 && !chain_forced)
            base_reads = std::max<uint32_t>(4, std::min<uint32_t>(chain_reads, (uint32_t)(units / ((uint64_t)kMixedRunsPerSlot * 2 * kNumSimd))));
        for (const auto &kv : by_shape)
            if (kv.second.chain)
                for (uint32_t g : kv.second.regions) {
                    double r = base_reads;
#ifdef PHMM_COST_SCALED_RUNS
                    if (!chain_forced) r = std::min<double>(CHAIN_MAX_READS, std::max(4.0, r * mean_cost / read_cost(g, kv.second.K) + 0.5));
#endif
                    reg_run[g] = std::min<uint32_t>(CHAIN_MAX_READS, (uint32_t)r * (uint32_t)kv.second.streams);
                }
    }
    // bytes of per-class work lists the plan will place in device memory
    size_t class_meta = 0;
    for (const auto &kv : by_shape) {
        const ShapeClass &c = kv.second;
        b->max_h = std::max(b->max_h, c.max_h);
        class_meta += align_up(c.reads.size() * 4, 256);
        if (c.chain) {
            const uint32_t gs = (uint32_t)(WAVE / c.L) / (uint32_t)c.streams;
            uint64_t items = 0;
            for (uint32_t g : c.regions)
                items += (uint64_t)((shape[g].nh + gs - 1) / gs + 3) * ((shape[g].nr + reg_run[g] - 1) / reg_run[g]);  // (+ 3: a remainder in items of its own, below)
            class_meta += align_up(items * sizeof(ChainItem), 256);
        }
        if (!c.L) class_meta += align_up((c.pair_first.size() + 1) * 8, 256);
    }
    mark();  // 3: classes, run lengths
    // ---- device memory provider (after planning: the arena is sized from the plan) -----------------
    // arena mode: bump-allocate from the handle's arena, "uploads" go to the pinned mirror and travel in
    // one copy later; otherwise hipMalloc per piece and async copies on the handle's stream.
    if (use_arena) {
        // exact: every dalloc() below and the payload / result placement of enqueue_compute (each piece starts on a
        // 256-byte boundary), so that nothing staged later can fail for lack of room
        const size_t need = align_up((size_t)n_reads * 4, 256) + align_up((size_t)(n_regions + 1) * 4, 256) * 2 +
                            align_up((size_t)(n_reads + 1) * 4, 256) + align_up((size_t)(n_haps + 1) * 4, 256) +
                            align_up((size_t)(n_regions + 1) * 8, 256) + class_meta + 5 * align_up(b->read_bytes, 256) +
                            align_up(b->hap_bytes, 256) + 256 + align_up(b->n_out * 8, 256) + 4096 + extra_arena_bytes +
                            align_up((size_t)n_reads, 256) /* redo flags of the f32-first mode */;
        Arena &A = h->A();
        if (!canary_before_staging(h, A)) return nullptr;  // (PHMM_MIRROR_CANARY: a store landed in the last call's result block after it returned)
        if (A.cap < need) {
            (void)hipStreamSynchronize(h->S());
            if (A.dev) (void)hipFree(A.dev);
            if (A.host) (void)hipHostFree(A.host);
            A.dev = A.host = nullptr;
            A.cap = 0;
            const size_t cap = std::max<size_t>(need + need / 2, 1 << 20);
            ok = hip_ok(h, hipMalloc((void **)&A.dev, cap), "hipMalloc(arena)") &&
                 hip_ok(h, hipHostMalloc((void **)&A.host, cap, hipHostMallocDefault), "hipHostMalloc(arena)");
            if (!ok) return nullptr;
            A.cap = cap;
        }
        A.used = 0;
        b->arena = &A;
    }
    auto dalloc = [&](size_t bytes, void **mirror) -> void * {
        if (mirror) *mirror = nullptr;
        if (!ok || dry) return nullptr;
        if (b->arena && align_up(b->arena->used, 256) + bytes <= b->arena->cap) {
            const size_t off = align_up(b->arena->used, 256);
            b->arena->used = off + bytes;
            if (mirror) *mirror = b->arena->host + off;
            return b->arena->dev + off;
        }
        void *p = nullptr;
        ok = hip_ok(h, hipMalloc(&p, std::max<size_t>(bytes, 256)), "hipMalloc(batch)");
        if (ok) b->mallocs.push_back(p);
        return p;
    };
    bool async_pending = false;
    auto up = [&](void *dst, void *mirror, const void *src, size_t bytes) {
        if (!ok || !bytes || dry) return;
        if (mirror) {
            memcpy(mirror, src, bytes);
        } else {
            ok = hip_ok(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->S()), "H2D meta");
            async_pending = true;
        }
    };

    // ---- device metadata --------------------------------------------------------------------
    void *m_rr, *m_rro, *m_rho, *m_ro, *m_ho, *m_oo;
    b->d_read_region = (uint32_t *)dalloc((size_t)n_reads * 4, &m_rr);
    b->d_region_read_off = (uint32_t *)dalloc((size_t)(n_regions + 1) * 4, &m_rro);
    b->d_region_hap_off = (uint32_t *)dalloc((size_t)(n_regions + 1) * 4, &m_rho);
    b->d_read_off = (uint32_t *)dalloc((size_t)(n_reads + 1) * 4, &m_ro);
    b->d_hap_off = (uint32_t *)dalloc((size_t)(n_haps + 1) * 4, &m_ho);
    b->d_out_off = (uint64_t *)dalloc((size_t)(n_regions + 1) * 8, &m_oo);
    up(b->d_read_region, m_rr, read_region.data(), (size_t)n_reads * 4);
    up(b->d_region_read_off, m_rro, region_read_off, (size_t)(n_regions + 1) * 4);
    up(b->d_region_hap_off, m_rho, region_hap_off, (size_t)(n_regions + 1) * 4);
    up(b->d_read_off, m_ro, read_off, (size_t)(n_reads + 1) * 4);
    up(b->d_hap_off, m_ho, hap_off, (size_t)(n_haps + 1) * 4);
    up(b->d_out_off, m_oo, out_off, (size_t)(n_regions + 1) * 8);
    if (!b->arena && !dry) {  // persistent batch: own status word (arena mode keeps it next to the results)
        b->d_status = (uint32_t *)dalloc(256, nullptr);
        if (ok) ok = hip_ok(h, hipMemsetAsync(b->d_status, 0, 4, h->S()), "memset status");
        // ... and own scratch for the exact pass, which rides behind the forward kernels of every launch (the caller
        // owns the stream, so the library cannot look at the status word in between)
        if (ok && n_reads && !sw.no_rescue) {
            size_t bytes = 0;
            rescue_geometry(b->max_h, &b->rescue_blocks, &bytes);
            ok = hip_ok(h, hipMalloc((void **)&b->rescue_scratch, bytes), "hipMalloc(rescue scratch)");
            if (ok) b->mallocs.push_back(b->rescue_scratch);
        }
    }

    mark();  // 4: arena, metadata
    // ---- finalise classes -------------------------------------------------------------------
    uint64_t best_cells = 0;
    for (auto &kv : by_shape) {
        ShapeClass c = std::move(kv.second);
        const uint32_t n_items = (uint32_t)c.reads.size();
        c.identity = (n_items == n_reads);
        for (uint32_t i = 0; c.identity && i < n_items; ++i) c.identity = (c.reads[i] == i);
        if (!c.identity && !c.chain) {
            void *mirror;
            c.d_reads = (uint32_t *)dalloc((size_t)n_items * 4, &mirror);
            up(c.d_reads, mirror, c.reads.data(), (size_t)n_items * 4);
        }
        if (c.chain) {
            for (uint32_t g : c.regions) {
                const uint32_t r0 = region_read_off[g], r1 = region_read_off[g + 1];
                const uint32_t gs = (uint32_t)(WAVE / c.L) / (uint32_t)c.streams;  // haplotypes per work item
                const uint32_t nq = (shape[g].nh + gs - 1) / gs;
                const uint32_t run = reg_run[g];
                // the haplotype groups of one run next to each other: they sweep the same read bytes, and items that are
                // launched together find them in L2 (config 3, 10 000 regions: HBM traffic 2.7 x the algorithmic bytes
                // with the groups a whole pass apart)
                // A haplotype count that leaves the last wave of a one-stream class partly empty (5 haplotypes: 4 + 1) gives
                // the remainder to items of its own with 2 or 4 streams of reads, which fill the wave's slots with the same
                // haplotypes again (chain_streams): 5 haplotypes 0.63 -> 0.96 of the slots busy, 9: 0.75 -> 0.98.
                uint32_t nq_main = nq, rest = 0, rest_streams = 1;
                if (c.streams == 1 && c.L == 16 && !(h->flags & PHMM_FLAG_F32_FIRST) && sw.force_streams == 0 && shape[g].nh > 4 && shape[g].nh % 4 != 0) {
                    rest = shape[g].nh % 4;
                    rest_streams = (uint32_t)chain_streams(rest, nullptr);
                    if (rest_streams > 1) nq_main = shape[g].nh / 4;
                    else rest = 0;
                }
                for (uint32_t r = r0; r < r1; r += run)
                    for (uint32_t q = 0; q < nq_main; ++q)
                        c.chain_items.push_back(ChainItem{g, (uint16_t)q, (uint8_t)c.K, (uint8_t)c.streams, r, std::min(r1, r + run)});
                if (rest) {
                    const uint32_t gs2 = 4 / rest_streams, q0 = nq_main * 4 / gs2, nq2 = (rest + gs2 - 1) / gs2;
                    const uint32_t run2 = std::min<uint32_t>(CHAIN_MAX_READS, run * rest_streams);
                    for (uint32_t r = r0; r < r1; r += run2)
                        for (uint32_t q = 0; q < nq2; ++q)
                            c.chain_items.push_back(ChainItem{g, (uint16_t)(q0 + q), (uint8_t)c.K, (uint8_t)rest_streams, r, std::min(r1, r + run2)});
                }
            }
            // (the launch is ordered longest item first below, across all classes: one sort there instead of one per class and
            // another over the whole -- the planner of a 186-region chunk of the ragged mix spent 1.3 of its 2.6 ms here)
            c.f32_first = (h->flags & PHMM_FLAG_F32_FIRST) && (c.L == 16 || c.L == 32);
            {   // the chained classes of one lanes-per-pair value (and one precision) share a launch
                phmm_batch::ChainGroup *grp = nullptr;
                for (auto &gq : b->chain_groups)
                    if (gq.L == c.L && gq.f32 == c.f32_first) grp = &gq;
                if (!grp) {
                    b->chain_groups.emplace_back();
                    grp = &b->chain_groups.back();
                    grp->L = c.L;
                    grp->f32 = c.f32_first;
                }
                grp->items.insert(grp->items.end(), c.chain_items.begin(), c.chain_items.end());
            }
            if (c.f32_first) {  // the f64 per-read kernel runs behind the f32 sweep over the reads it flags
                if (!c.identity) {
                    void *mr;
                    c.d_reads = (uint32_t *)dalloc((size_t)n_items * 4, &mr);
                    up(c.d_reads, mr, c.reads.data(), (size_t)n_items * 4);
                }
                c.lds_rows = (uint32_t)align_up((size_t)c.max_r + 1, 8);
                c.waves_per_block = 1;
                c.lds_bytes = (size_t)c.lds_rows * kLdsRowBytes;
                c.grid = dim3(n_items, 1, 1);  // one wave per read, it walks all haplotype groups
                c.cnd_select = 0;
                if (!b->d_redo) b->d_redo = (uint8_t *)dalloc(align_up((size_t)n_reads, 256), nullptr);
            }
            const char *f32 = c.f32_first ? "_f32" : "";
            if (c.streams > 1)
                snprintf(c.name, sizeof c.name, "phmm_forward_chain%s<%d,%d> x%d streams", f32, c.L, c.K, c.streams);
            else
                snprintf(c.name, sizeof c.name, "phmm_forward_chain%s<%d,%d>", f32, c.L, c.K);
        } else if (c.L) {
            c.lds_rows = (uint32_t)align_up((size_t)c.max_r + 1, 8);
            const size_t per_wave = (size_t)c.lds_rows * kLdsRowBytes;
            // One wave per workgroup: waves are independent (no barrier, private LDS), and a multi-wave block
            // would hold its LDS until its longest read finishes -- with mixed read lengths that idles SIMDs.
            c.waves_per_block = 1;
            c.lds_bytes = per_wave * c.waves_per_block;
            // Enough reads to fill the chip -> one wave walks all haplotype groups of its read (row
            // constants staged once); otherwise spread the groups over gridDim.y.
            bool split = (uint64_t)n_items < 4ull * kNumSimd;
            c.grid = dim3((n_items + c.waves_per_block - 1) / c.waves_per_block, split ? c.max_quads : 1, 1);
            // a wave alone on its SIMD is latency-bound: the v_cndmask select (one more VALU op, no EXEC round
            // trip) is ~8 % faster there; with two resident waves the EXEC-masked select wins
            const uint64_t waves = (uint64_t)n_items * (split ? c.max_quads : 1);
            c.cnd_select = waves < 2ull * kNumSimd ? 1u : 0u;
            snprintf(c.name, sizeof c.name, "phmm_forward<%d,%d>", c.L, c.K);
        } else {
            // generic: exclusive prefix of pairs per read, scratch for a bounded grid
            uint64_t acc = 0;
            for (auto &v : c.pair_first) {
                const uint64_t nh = v;
                v = acc;
                acc += nh;
            }
            c.pair_first.push_back(acc);
            const uint64_t per_thread = 6ull * (c.max_h + 1) * sizeof(double);
            uint64_t threads = std::min<uint64_t>(align_up(acc, 256), 1024ull * 256);
            threads = std::min<uint64_t>(threads, std::max<uint64_t>(256, kGenericScratchBytes / per_thread / 256 * 256));
            c.generic_blocks = (uint32_t)(threads / 256);
            // scratch can be large: always its own allocation, never the arena
            if (ok && !dry) ok = hip_ok(h, hipMalloc((void **)&c.d_scratch, threads * per_thread), "hipMalloc(generic scratch)");
            if (ok && !dry) b->mallocs.push_back(c.d_scratch);
            void *mirror;
            c.d_pair_first = (uint64_t *)dalloc(c.pair_first.size() * 8, &mirror);
            up(c.d_pair_first, mirror, c.pair_first.data(), c.pair_first.size() * 8);
            snprintf(c.name, sizeof c.name, "phmm_forward_generic");
        }
        if (c.cells >= best_cells) {
            best_cells = c.cells;
            b->dominant = c.name;
        }
        if (sw.trace)
            fprintf(stderr, "  class %-40s regions %6zu reads %8zu items %8zu cells %.3e max_h %u\n", c.name, c.regions.size(),
                    c.reads.size(), c.chain_items.size(), (double)c.cells, c.max_h);
        b->classes.push_back(std::move(c));
    }
    mark();  // 5: work items per class
    {   // What these launches sweep, padding and all (phmm_batch_executed_cells), in lane-cells = steps x 64 lanes x K columns per
        // wave, and where the padding comes from: columns beyond a haplotype's end (16 K - H), haplotype slots a wave leaves
        // empty, and steps that carry no read row (the SUM / RESET rows between the reads of a run, the L - 1 steps a run needs to
        // reach its last lane, the rows the longest of a wave's streams has more than the others).
        uint64_t swept = 0, pad_cols = 0, pad_slots = 0, t_marks = 0, t_fill = 0, t_uneven = 0, t_uneven_best = 0;  // (t_*: PHMM_TRACE only)
        auto haps_of = [&](uint32_t g, uint32_t first, uint32_t slots, uint32_t lanes_cols, uint64_t &sum_h, uint32_t &valid) {
            const uint32_t h0 = region_hap_off[g], nh = region_hap_off[g + 1] - h0;
            sum_h = 0;
            valid = 0;
            for (uint32_t a = first; a < first + slots && a < nh; ++a) {
                sum_h += std::min<uint32_t>(hap_off[h0 + a + 1] - hap_off[h0 + a], lanes_cols);
                ++valid;
            }
        };
        for (const auto &grp : b->chain_groups)
            for (const ChainItem &x : grp.items) {
                const uint32_t S = std::max<uint32_t>(1, x.streams), L = (uint32_t)grp.L, GS = (64u / L) / S, LK = L * x.k;
                const uint32_t n = x.read_end - x.read_begin, n_sub = (n + S - 1) / S;
                uint64_t longest = 0, read_rows = 0;
                for (uint32_t st = 0; st < S; ++st) {  // (stream st sweeps reads [st n_sub, (st + 1) n_sub) of the run)
                    const uint32_t lo = std::min(n, st * n_sub), hi = std::min(n, lo + n_sub);
                    const uint64_t rows = read_off[x.read_begin + hi] - read_off[x.read_begin + lo];
                    read_rows += rows;
                    longest = std::max<uint64_t>(longest, rows + 2ull * (hi - lo));
                }
                const uint64_t steps = (longest + L) & ~1ull;
                swept += steps * 64ull * x.k;
                t_marks += 2ull * n * GS * LK;                                          // the SUM / RESET rows of its reads
                t_fill += (steps - longest) * 64ull * x.k;                              // reaching the last lane
                t_uneven += (longest * S - read_rows - 2ull * n) * (uint64_t)GS * LK;   // streams shorter than the longest
                if (sw.trace && S > 1) {  // ... and what the best cut of the run into S contiguous parts would leave of that
                    uint64_t lo_b = 0, hi_b = read_rows + 2ull * n;
                    for (uint32_t i = 0; i < n; ++i) lo_b = std::max<uint64_t>(lo_b, read_off[x.read_begin + i + 1] - read_off[x.read_begin + i] + 2);
                    while (lo_b < hi_b) {
                        const uint64_t mid = (lo_b + hi_b) / 2;
                        uint32_t parts = 1;
                        uint64_t acc = 0;
                        for (uint32_t i = 0; i < n; ++i) {
                            const uint64_t len = read_off[x.read_begin + i + 1] - read_off[x.read_begin + i] + 2;
                            if (acc + len > mid) {
                                ++parts;
                                acc = 0;
                            }
                            acc += len;
                        }
                        if (parts <= S) hi_b = mid; else lo_b = mid + 1;
                    }
                    t_uneven_best += (lo_b * S - read_rows - 2ull * n) * (uint64_t)GS * LK;
                }
                uint64_t sum_h;
                uint32_t valid;
                haps_of(x.region, (uint32_t)x.quad * GS, GS, LK, sum_h, valid);
                pad_cols += read_rows * ((uint64_t)valid * LK - sum_h);
                pad_slots += read_rows * (uint64_t)(GS - valid) * LK;
            }
        for (const auto &c : b->classes) {
            if (c.chain) continue;  // (counted above; the f64 redo behind an f32 sweep touches the reads it flags only)
            if (!c.L) {
                swept += c.cells;
                continue;
            }
            const size_t n = c.identity ? n_reads : c.reads.size();
            const uint32_t per_wave = 64u / (uint32_t)c.L, LK = (uint32_t)(c.L * c.K);
            for (size_t i = 0; i < n; ++i) {
                const uint32_t r = c.identity ? (uint32_t)i : c.reads[i], g = read_region[r];
                const uint32_t quads = (shape[g].nh + per_wave - 1) / per_wave;
                const uint64_t rows = read_off[r + 1] - read_off[r];
                swept += (rows + (uint64_t)c.L - 1) * quads * 64ull * (uint64_t)c.K;
                for (uint32_t qd = 0; qd < quads; ++qd) {
                    uint64_t sum_h;
                    uint32_t valid;
                    haps_of(g, qd * per_wave, per_wave, LK, sum_h, valid);
                    pad_cols += rows * ((uint64_t)valid * LK - sum_h);
                    pad_slots += rows * (uint64_t)(per_wave - valid) * LK;
                }
            }
        }
        if (sw.trace)
            fprintf(stderr, "phmm plan: swept %.4e lane-cells for %.4e cells: columns %.4e, slots %.4e; chained items' SUM / RESET rows %.4e, fill %.4e, uneven streams %.4e (cut by rows: %.4e)\n",
                    (double)swept, (double)b->cells, (double)pad_cols, (double)pad_slots, (double)t_marks, (double)t_fill, (double)t_uneven, (double)t_uneven_best);
        b->swept_cells = swept;
        b->pad_column_cells = pad_cols;
        b->pad_slot_cells = pad_slots;
    }
    for (auto &grp : b->chain_groups) {
        // longest item first across all classes of the launch: (rows of the run + its SUM / RESET rows) x the cost of a
        // step at the item's K (7 VALU per column + ~11 per step)
        auto cost = [&](const ChainItem &x) {
            return (uint64_t)(read_off[x.read_end] - read_off[x.read_begin] + 2 * (x.read_end - x.read_begin) + grp.L) *
                   (uint64_t)(7 * x.k + 11);
        };
        {   // (keys made once -- the comparator used to fetch four offsets per comparison -- and unique, so a plain sort keeps
            // items of equal cost in the order they were made: the groups of a run stay next to each other)
            const size_t n = grp.items.size();
            std::vector<uint64_t> cst(n);
            uint64_t top = 0;
            for (size_t i = 0; i < n; ++i) top = std::max(top, cst[i] = cost(grp.items[i]));
            std::vector<uint32_t> idx(n), tmp(n);
            for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
            if (top < (1ull << 33)) {  // LSD radix sort, descending, 11 bits a pass (stable): tens of microseconds for 10^4 items
                for (int shift = 0; (top >> shift) != 0; shift += 11) {
                    uint32_t count[2049] = {0};
                    for (size_t i = 0; i < n; ++i) count[2047 - ((cst[idx[i]] >> shift) & 2047) + 1] += 1;
                    for (int d = 0; d < 2048; ++d) count[d + 1] += count[d];
                    for (size_t i = 0; i < n; ++i) tmp[count[2047 - ((cst[idx[i]] >> shift) & 2047)]++] = idx[i];
                    idx.swap(tmp);
                }
            } else {
                std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return cst[x] > cst[y]; });
            }
            std::vector<ChainItem> sorted(n);
            for (size_t i = 0; i < n; ++i) sorted[i] = grp.items[idx[i]];
            grp.items.swap(sorted);
        }
        // XCD-aware placement.  The haplotype groups of one run (same region, same reads: equal cost, so the stable sort
        // left them next to each other) sweep the same read bytes.  Workgroups are dealt to the eight XCDs round robin,
        // each XCD with an L2 of its own, so neighbours in the launch never share one: take eight runs at a time and
        // emit their first groups, then their second groups, ... -- the groups of a run are then 8 blocks apart, on
        // the same XCD, started together.
        {
            std::vector<ChainItem> out;
            out.reserve(grp.items.size());
            auto same_run = [](const ChainItem &x, const ChainItem &y) {
                return x.region == y.region && x.read_begin == y.read_begin && x.read_end == y.read_end;
            };
            size_t i = 0;
            const size_t n = grp.items.size();
            while (i < n) {
                size_t start[9], len[8];  // up to eight consecutive runs
                int nr = 0;
                size_t j = i;
                while (nr < 8 && j < n) {
                    size_t e = j + 1;
                    while (e < n && same_run(grp.items[j], grp.items[e])) ++e;
                    start[nr] = j;
                    len[nr] = e - j;
                    ++nr;
                    j = e;
                }
                size_t longest = 0;
                for (int r = 0; r < nr; ++r) longest = std::max(longest, len[r]);
                for (size_t q = 0; q < longest; ++q)
                    for (int r = 0; r < nr; ++r)
                        if (q < len[r]) out.push_back(grp.items[start[r] + q]);
                i = j;
            }
            grp.items.swap(out);
        }
        grp.single_k = grp.items.empty() ? 0 : grp.items[0].k;
        for (const ChainItem &it : grp.items)
            if (it.k != grp.single_k) {
                grp.single_k = 0;
                break;
            }
    }
    {   // A mixed f64 group goes out as one launch per RANGE of K (the kernel of a range holds only its bodies: no spilled
        // scalar registers, no scratch); the launches of a batch run side by side on parallel streams (phmm_batch_launch).
        std::vector<phmm_batch::ChainGroup> split;
        for (auto &grp : b->chain_groups) {
            if (grp.f32 || grp.single_k != 0 || grp.items.empty()) {
                split.push_back(std::move(grp));
                continue;
            }
            phmm_batch::ChainGroup part[kChainRanges];
            for (const ChainItem &it : grp.items) part[chain_range_of(it.k)].items.push_back(it);  // (order kept: longest first)
            for (int r = 0; r < kChainRanges; ++r) {
                if (part[r].items.empty()) continue;
                part[r].L = grp.L;
                part[r].f32 = false;
                part[r].single_k = part[r].items[0].k;
                for (const ChainItem &it : part[r].items)
                    if (it.k != part[r].single_k) {
                        part[r].single_k = -(r + 1);
                        break;
                    }
                split.push_back(std::move(part[r]));
            }
        }
        // the heaviest launch first (it starts on the caller's stream, the others join it from the side streams)
        auto weight = [&](const phmm_batch::ChainGroup &g) {
            uint64_t w = 0;
            for (const ChainItem &x : g.items) w += (uint64_t)(read_off[x.read_end] - read_off[x.read_begin]) * (uint64_t)(7 * x.k + 11);
            return w;
        };
        std::stable_sort(split.begin(), split.end(), [&](const phmm_batch::ChainGroup &x, const phmm_batch::ChainGroup &y) { return weight(x) > weight(y); });
        b->chain_groups.swap(split);
    }
    // the dominant class under the name of the kernel that runs it (what rocprofv3 reports): the body alone for a launch
    // whose items share one K, the kernel of its range of K otherwise
    for (const auto &c : b->classes) {
        if (!c.chain || b->dominant != c.name) continue;
        for (const auto &grp : b->chain_groups) {
            if (grp.L != c.L || grp.f32 != c.f32_first) continue;
            char nm[64] = {0};
            if (grp.f32) {
                if (grp.single_k == c.K) snprintf(nm, sizeof nm, "phmm_forward_chain_f32<%d,%d>", c.L, c.K);
                else if (grp.single_k == 0) snprintf(nm, sizeof nm, "phmm_forward_chain_f32_any<%d> (K = %d)", c.L, c.K);
            } else if (grp.single_k == c.K) {
                snprintf(nm, sizeof nm, "phmm_forward_chain_k<%d,%d>", c.L, c.K);
            } else if (grp.single_k < 0 && chain_range_of(c.K) == -grp.single_k - 1) {
#define PHMM_RANGE(R, LO, HI) \
    if (R == -grp.single_k - 1) snprintf(nm, sizeof nm, "phmm_forward_chain<%d,%d,%d> (K = %d)", c.L, LO, HI, c.K);
                PHMM_CHAIN_RANGES(PHMM_RANGE)
#undef PHMM_RANGE
            }
            if (nm[0]) {
                b->dominant = nm;
                if (c.streams > 1) b->dominant += " x" + std::to_string(c.streams) + " streams";
                break;
            }
        }
        break;
    }
    mark();  // 6: sorting, placement, ranges
    if (sw.trace)
        fprintf(stderr, "  plan phases (us): shapes %.0f, <L,K> %.0f, classes %.0f, metadata %.0f, items %.0f, order %.0f\n", marks[1] - marks[0],
                marks[2] - marks[1], marks[3] - marks[2], marks[4] - marks[3], marks[5] - marks[4], marks[6] - marks[5]);
    for (auto &grp : b->chain_groups) {
        void *mirror;
        grp.d_items = (ChainItem *)dalloc(grp.items.size() * sizeof(ChainItem), &mirror);
        up(grp.d_items, mirror, grp.items.data(), grp.items.size() * sizeof(ChainItem));
    }
    // host staging vectors die at return: finish the async copies first (arena mode copied into the mirror)
    if (ok && async_pending) ok = hip_ok(h, hipStreamSynchronize(h->S()), "sync(meta)");
    if (!ok) return nullptr;
    return owner.release();
}

extern "C" {

phmm_batch *phmm_batch_create(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off,
                              const uint32_t *region_hap_off, const uint32_t *read_off, const uint32_t *hap_off,
                              const uint64_t *out_off) {
    PHMM_GUARD_BEGIN
    return batch_create_impl(h, n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, false);
    PHMM_GUARD_END(h, "phmm_batch_create", PHMM_FAIL_NULL)
}

// What the planned launches sweep, in lane-cells: every row of every wave x 64 lanes x its K columns -- the columns beyond a
// haplotype's end inside its 16 x K lanes, haplotype slots a wave leaves empty, the rows of a multi-stream item's shorter
// streams.  executed / cells is the padding a batch's shapes cost (1.01 for the uniform config-2 batch: 304 columns for 300).
uint64_t phmm_batch_executed_cells(const phmm_batch *b) { return b ? b->swept_cells : 0; }

int phmm_batch_bind_device(phmm_batch *b, const uint8_t *d_read_bases, const uint8_t *d_base_q, const uint8_t *d_ins_q,
                           const uint8_t *d_del_q, const uint8_t *d_gcp, const uint8_t *d_hap_bases, double *d_out) {
    if (!b) return PHMM_ERR_INVALID_ARG;
    if ((b->read_bytes && (!d_read_bases || !d_base_q || !d_ins_q || !d_del_q || !d_gcp)) ||
        (b->hap_bytes && !d_hap_bases) || (b->n_out && !d_out)) {
        b->h->err = "phmm_batch_bind_device: null device pointer";
        return PHMM_ERR_INVALID_ARG;
    }
    b->d_read_bases = d_read_bases;
    b->d_base_q = d_base_q;
    b->d_ins_q = d_ins_q;
    b->d_del_q = d_del_q;
    b->d_gcp = d_gcp;
    b->d_hap_bases = d_hap_bases;
    b->d_out = d_out;
    b->bound = true;
    return PHMM_OK;
}

int phmm_batch_upload(phmm_batch *b, const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q,
                      const uint8_t *del_q, const uint8_t *gcp, const uint8_t *hap_bases) {
    if (!b) return PHMM_ERR_INVALID_ARG;
    phmm_handle *h = b->h;
    if ((b->read_bytes && (!read_bases || !base_q || !ins_q || !del_q || !gcp)) || (b->hap_bytes && !hap_bases)) {
        h->err = "phmm_batch_upload: null host pointer";
        return PHMM_ERR_INVALID_ARG;
    }
    DeviceGuard dg(h->device);
    const size_t rb = align_up(b->read_bytes, 256), hb = align_up(b->hap_bytes, 256), ob = align_up(b->n_out * 8, 256);
    if (!b->d_owned) {
        HIP_TRY(h, hipMalloc(&b->d_owned, 5 * rb + hb + ob + 256), PHMM_ERR_HIP);
        b->mallocs.push_back(b->d_owned);
    }
    uint8_t *p = (uint8_t *)b->d_owned;
    uint8_t *d[6];
    for (int i = 0; i < 5; ++i) { d[i] = p; p += rb; }
    d[5] = p; p += hb;
    double *d_out = (double *)p;
    const uint8_t *src[6] = {read_bases, base_q, ins_q, del_q, gcp, hap_bases};
    for (int i = 0; i < 6; ++i) {
        const size_t bytes = i < 5 ? b->read_bytes : b->hap_bytes;
        if (bytes) HIP_TRY(h, hipMemcpyAsync(d[i], src[i], bytes, hipMemcpyHostToDevice, b->home_stream), PHMM_ERR_HIP);
    }
    return phmm_batch_bind_device(b, d[0], d[1], d[2], d[3], d[4], d[5], d_out);
}

// Everything of a launch that does not depend on the shape class.
static ForwardParams base_params(const phmm_batch *b) {
    const phmm_handle *h = b->h;
    ForwardParams p{};
    p.read_region = b->d_read_region;
    p.region_read_off = b->d_region_read_off;
    p.region_hap_off = b->d_region_hap_off;
    p.read_off = b->d_read_off;
    p.hap_off = b->d_hap_off;
    p.out_off = b->d_out_off;
    p.read_bases = b->d_read_bases;
    p.base_q = b->d_base_q;
    p.ins_q = b->d_ins_q;
    p.del_q = b->d_del_q;
    p.gcp = b->d_gcp;
    p.hap_bases = b->d_hap_bases;
    p.out = b->d_out;
    p.eps = h->d_eps;
    p.eps_mis = h->d_eps_mis;
    p.mm = h->d_mm;
    p.ratio_mis = h->d_ratio_mis;
    p.inv_om = h->d_inv_om;
    p.initial_condition = initial_condition();
    p.initial_condition_log10 = initial_condition_log10();
    p.status = b->d_status;
    return p;
}

// The exact pass over the pairs below kRescueBelow (phmm_exact_kernels.hip).  force == 0: in-stream, the kernel looks
// at the status word itself and returns at once when no forward kernel raised STATUS_RESCUE.
static int launch_rescue_pass(phmm_batch *b, double *scratch, uint32_t n_blocks, bool force, hipStream_t stream) {
    RescueParams rp{};
    rp.f = base_params(b);
    rp.n_reads = b->n_reads;
    rp.scratch = scratch;
    rp.max_h = b->max_h;
    rp.n_blocks = n_blocks;
    rp.force = force ? 1u : 0u;
    return hip_ok(b->h, launch_rescue(rp, stream), "phmm_rescue") ? PHMM_OK : PHMM_ERR_HIP;
}

// Scratch of the exact pass for batches staged in arena `A` (grow-only; the caller has made sure nothing that could
// use the old buffer is in flight on this slot).
static bool ensure_arena_rescue(phmm_handle *h, Arena &A, uint32_t max_h, uint32_t *n_blocks) {
    size_t bytes = 0;
    rescue_geometry(max_h, n_blocks, &bytes);
    if (A.rescue_cap >= bytes) return true;
    if (A.rescue) (void)hipFree(A.rescue);
    A.rescue = nullptr;
    A.rescue_cap = 0;
    if (!hip_ok(h, hipMalloc((void **)&A.rescue, bytes), "hipMalloc(rescue scratch)")) return false;
    A.rescue_cap = bytes;
    return true;
}

int phmm_batch_launch(phmm_batch *b, void *stream_v) {
    if (!b) return PHMM_ERR_INVALID_ARG;
    phmm_handle *h = b->h;
    if (!b->bound) {
        h->err = "phmm_batch_launch: no device buffers bound";
        return PHMM_ERR_NOT_BOUND;
    }
    DeviceGuard dg(h->device);
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : b->home_stream;
    if (b->d_redo && !hip_ok(h, hipMemsetAsync(b->d_redo, 0, b->n_reads, stream), "memset redo")) return PHMM_ERR_HIP;
    // The chained sweeps: one launch per lanes-per-pair value, precision and (mixed batches) range of K.  Several launches
    // run side by side: the first on the caller's stream, the others on the handle's side streams between a fork and a
    // join event -- each alone would leave the chip to its own tail before the next could start.
    const size_t n_groups = b->chain_groups.size();
    // (not while the chunks of a pipelined host call are in flight: those already overlap each other on the slot streams,
    // and forks of several chunks would queue behind one another on the side streams -- 1 536 mixed regions through host
    // buffers: 25 ms without, 32 ms with)
    // (round 4, three large chunks: forking them all changes nothing either; round 5, the LAST chunk of a mixed call alone: 19.8-20.0
    // ms against 18.7-18.8 -- both measured and dropped)
    const bool fork = n_groups >= 2 && !h->defer_d2h;
    if (fork) {
        for (int i = 0; i < phmm_handle::kSideStreams; ++i) {
            if (!h->side_streams[i] && !hip_ok(h, hipStreamCreateWithFlags(&h->side_streams[i], hipStreamNonBlocking), "hipStreamCreate")) return PHMM_ERR_HIP;
            if (!h->ev_join[i] && !hip_ok(h, hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming), "hipEventCreate")) return PHMM_ERR_HIP;
        }
        if (!h->ev_fork && !hip_ok(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming), "hipEventCreate")) return PHMM_ERR_HIP;
        if (!hip_ok(h, hipEventRecord(h->ev_fork, stream), "hipEventRecord")) return PHMM_ERR_HIP;
    }
    auto launch_class = [&](ShapeClass &c) -> int {
        ForwardParams p = base_params(b);
        p.class_reads = c.identity ? nullptr : c.d_reads;
        p.n_items = (uint32_t)c.reads.size();
        p.lds_rows = c.lds_rows;
        p.cnd_select = c.cnd_select;
        if (!p.n_items) return PHMM_OK;
        hipError_t e;
        if (c.chain && !c.f32_first) return PHMM_OK;  // done with its group
        if (c.chain) {  // behind the f32 sweep: the f64 per-read kernel over exactly the reads it flagged
            p.redo = b->d_redo;
            e = launch_forward(c.L, c.K, p, c.grid, c.waves_per_block, c.lds_bytes, stream);
        } else if (c.L) {
            e = launch_forward(c.L, c.K, p, c.grid, c.waves_per_block, c.lds_bytes, stream);
        } else {
            GenericParams gp{};
            gp.f = p;
            gp.scratch = c.d_scratch;
            gp.max_h = c.max_h;
            gp.pair_first = c.d_pair_first;
            gp.n_pairs = c.pair_first.back();
            gp.n_blocks = c.generic_blocks;
            e = gp.n_pairs ? launch_generic(gp, stream) : hipSuccess;
        }
        return hip_ok(h, e, c.name) ? PHMM_OK : PHMM_ERR_HIP;
    };
    // The per-read classes of a mixed batch (a few hundred microseconds of small kernels) depend on no chained launch: behind
    // the join they ran alone at the end of the batch, one after the other.  With the launches forked they go out FIRST on the
    // caller's stream -- the chained launches of the side streams start beside them, the caller's own behind them.
    // (The mixed batch resident, three runs each on one box: 15.59-15.66 ms against 15.80-15.89 the other way round.)
    const bool early_classes = fork;
    if (early_classes)
        for (auto &c : b->classes)
            if (!c.chain && launch_class(c) != PHMM_OK) return PHMM_ERR_HIP;
    bool side_used[phmm_handle::kSideStreams] = {};
    // (Measured and dropped, round 5: dealing the launches to the streams by load, every stream sending its lightest first -- in the
    // mixed batch's timeline the two lightest launches trail the second and third heaviest -- : 15.56-15.67 ms either way.)
    for (size_t gi = 0; gi < n_groups; ++gi) {
        hipStream_t s_x = stream;
        if (fork && gi > 0) {
            const int si = (int)((gi - 1) % phmm_handle::kSideStreams);
            s_x = h->side_streams[si];
            if (!side_used[si] && !hip_ok(h, hipStreamWaitEvent(s_x, h->ev_fork, 0), "hipStreamWaitEvent")) return PHMM_ERR_HIP;
            side_used[si] = true;
        }
        auto &grp = b->chain_groups[gi];
        if (grp.items.empty()) continue;
        ChainParams cp{};
        cp.f = base_params(b);
        cp.items = grp.d_items;
        cp.n_items = (uint32_t)grp.items.size();
        cp.redo = grp.f32 ? b->d_redo : nullptr;
        hipStream_t s = s_x;
        const hipError_t e = grp.f32 ? launch_chain_f32(grp.L, grp.single_k, cp, s) : launch_chain(grp.L, grp.single_k, cp, s);
        if (!hip_ok(h, e, grp.f32 ? "phmm_forward_chain_f32" : "phmm_forward_chain")) return PHMM_ERR_HIP;
    }
    for (int i = 0; i < phmm_handle::kSideStreams; ++i)
        if (side_used[i] && (!hip_ok(h, hipEventRecord(h->ev_join[i], h->side_streams[i]), "hipEventRecord") ||
                             !hip_ok(h, hipStreamWaitEvent(stream, h->ev_join[i], 0), "hipStreamWaitEvent")))
            return PHMM_ERR_HIP;
    // (the per-read classes that depend on no chained launch went out in front of the fork, above)
    for (auto &c : b->classes)
        if (!(early_classes && !c.chain) && launch_class(c) != PHMM_OK) return PHMM_ERR_HIP;
    // Results below kRescueBelow are redone in the reference's operation order.  Persistent batches and the
    // engine-level call carry the pass in-stream (it returns at once unless a forward kernel asked for it); the
    // host-buffer path looks at the status word in finish_compute instead and pays nothing in the common case.
    if (b->rescue_scratch && !h->sw.no_rescue && b->n_reads)
        return launch_rescue_pass(b, b->rescue_scratch, b->rescue_blocks, false, stream);
    return PHMM_OK;
}

int phmm_batch_status(phmm_batch *b) {
    if (!b) return PHMM_ERR_INVALID_ARG;
    phmm_handle *h = b->h;
    DeviceGuard dg(h->device);
    uint32_t st = 0;
    HIP_TRY(h, hipMemcpy(&st, b->d_status, 4, hipMemcpyDeviceToHost), PHMM_ERR_HIP);
    if (st) HIP_TRY(h, hipMemset(b->d_status, 0, 4), PHMM_ERR_HIP);
    if (status_positive(st, b->rescue_scratch != nullptr && !h->sw.no_rescue)) {  // (the exact pass rode behind every launch: its verdict)
        h->err = "PairHmm Log Probability cannot be greater than 0.0";  // pair_hmm.rs:478-481
        return PHMM_ERR_POSITIVE_RESULT;
    }
    return PHMM_OK;
}

int phmm_batch_download(phmm_batch *b, double *out) {
    if (!b || (b->n_out && !out)) return PHMM_ERR_INVALID_ARG;
    phmm_handle *h = b->h;
    if (!b->bound) return PHMM_ERR_NOT_BOUND;
    DeviceGuard dg(h->device);
    if (b->n_out)
        HIP_TRY(h, hipMemcpyAsync(out, b->d_out, b->n_out * 8, hipMemcpyDeviceToHost, b->home_stream), PHMM_ERR_HIP);
    HIP_TRY(h, hipStreamSynchronize(b->home_stream), PHMM_ERR_HIP);
    return phmm_batch_status(b);
}

}  // extern "C"

namespace phmm_host {

// When is the D2H copy of the results enqueued?  Right behind the kernels ("eager": one host wait per call, 15 us less
// latency for a lone small call), or by finish_compute once the host has seen the kernels complete ("deferred").  A copy
// that waits for a kernel sits at the head of its SDMA engine's queue and holds up every later copy that lands on that
// engine -- the H2D of the next chunk, or of another lane (tools/ubench/overlap3.hip; seen as strict H2D / kernel / D2H
// serialisation in the rocprofv3 timeline of the chunked path).  So: deferred wherever something else is in flight
// (chunks of a pipelined call, combined flushes of phmm_wait), eager for a one-shot call.
// PHMM_EAGER_D2H=1 / 0 forces one or the other (A/B measurements only).
bool eager_d2h(const phmm_handle *h) {
    return kForcedEagerD2H >= 0 ? kForcedEagerD2H != 0 : !h->defer_d2h;
}

// The zero-copy path (inputs fetched from the pinned mirror by a kernel, results stored into it) has no copies to keep out of
// each other's way: a combined flush may take it too.  (PHMM_EAGER_D2H=0 still forces the copy path, for the tests.)
bool zero_copy_allowed(const phmm_handle *) { return kForcedEagerD2H >= 0 ? kForcedEagerD2H != 0 : true; }

// ---- PHMM_MIRROR_CANARY ------------------------------------------------------------------------------------------------
// Small calls hand inputs and results over through the pinned mirror, and their kernels tell the calling thread themselves
// when they are through (region_finish): nothing but the kernels' own ordering stands between a late store and the next
// call's staged inputs.  Round 4 shipped such a store (two timing words of the aligner, NOTEBOOK 18.7) that only a C++
// caller saw, as one read's likelihoods 20 decades low.  With the switch set every such store is a failed call:
//   * when a zero-copy call returns, its result block in the mirror is filled with 0xA5 and must still be 0xA5 when the arena
//     is staged again (a store that lands between two calls);
//   * the inputs a zero-copy call staged are kept aside and compared with the mirror when the call ends (a store that lands
//     in the NEXT call's inputs -- the device never writes there).
static bool canary_fail(phmm_handle *h, const char *what, size_t at, unsigned got) {
    char msg[256];
    snprintf(msg, sizeof msg, "PHMM_MIRROR_CANARY: %s: pinned mirror offset %zu holds 0x%02x", what, at, got);
    fprintf(stderr, "%s\n", msg);
    if (h->sw.mirror_canary >= 2) abort();
    h->err = msg;
    h->err_code = PHMM_ERR_INTERNAL;
    return false;
}
bool canary_before_staging(phmm_handle *h, Arena &A) {
    A.canary_inputs.clear();
    if (!h->sw.mirror_canary || !A.canary_bytes || !A.host) return true;
    const size_t off = A.canary_off, n = A.canary_bytes;
    A.canary_bytes = 0;
    if (off + n > A.cap) return true;  // (the arena was replaced by a larger one)
    const unsigned char *q = (const unsigned char *)A.host + off;
    for (size_t i = 0; i < n; ++i)
        if (q[i] != 0xA5) return canary_fail(h, "a device store landed in a result block after its call had returned", off + i, q[i]);
    return true;
}
void canary_staged(phmm_handle *h, Arena &A, size_t in_bytes) {
    if (!h->sw.mirror_canary) return;
    A.canary_inputs.assign((const unsigned char *)A.host, (const unsigned char *)A.host + in_bytes);
}
bool canary_after_call(phmm_handle *h, Arena &A, size_t res_off, size_t res_bytes) {
    if (!h->sw.mirror_canary) return true;
    bool good = true;
    const size_t n = std::min(A.canary_inputs.size(), A.cap);
    if (n && memcmp(A.host, A.canary_inputs.data(), n) != 0) {
        size_t i = 0;
        while (i < n && (unsigned char)A.host[i] == A.canary_inputs[i]) ++i;
        good = canary_fail(h, "a device store landed in the staged inputs of a call", i, (unsigned char)A.host[i]);
    }
    A.canary_inputs.clear();
    if (res_off + res_bytes <= A.cap) {
        memset(A.host + res_off, 0xA5, res_bytes);
        A.canary_off = res_off;
        A.canary_bytes = res_bytes;
    }
    return good;
}

// Stage one batch in the current slot's arena and enqueue H2D, kernels and D2H on its stream.  No sync.
int enqueue_compute(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                    const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q,
                    const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off, const uint8_t *hap_bases,
                    const uint64_t *out_off, double *out, PendingCompute *pending, const Parts *parts) {
    const bool trace = h->sw.trace != 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = now();
    double t_plan = 0, t_stage = 0, t_h2d = 0, t_launch = 0;
    bool zero_copy = false;
    phmm_batch *b = batch_create_impl(h, n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, true);
    if (!b) return h->err_code ? h->err_code : PHMM_ERR_INVALID_ARG;
    int st = PHMM_OK;
    if (!parts && ((b->read_bytes && (!read_bases || !base_q || !ins_q || !del_q || !gcp)) || (b->hap_bytes && !hap_bases) ||
                   (b->n_out && !out))) {
        h->err = "phmm_compute: null pointer";
        st = PHMM_ERR_INVALID_ARG;
    }
    Arena &A = h->A();
    t_plan = now();
    if (st == PHMM_OK) {
        // payload into the arena mirror, then [status | out] last so that one copy each way suffices
        const uint8_t *src[6] = {read_bases, base_q, ins_q, del_q, gcp, hap_bases};
        const uint8_t *d[6];
        size_t offs[6] = {0};
        for (int i = 0; i < 6; ++i) {
            const size_t bytes = i < 5 ? b->read_bytes : b->hap_bytes;
            const size_t off = align_up(A.used, 256);
            A.used = off + bytes;
            if (A.used > A.cap) {
                h->err = "phmm_compute: internal error, arena too small";
                st = PHMM_ERR_HIP;
                break;
            }
            offs[i] = off;
            h->stat_staged_bytes += bytes;
            d[i] = (const uint8_t *)(A.dev + off);
        }
        // everything rides in the single copy of the pinned mirror (the chunked path keeps every array of a
        // chunk at or below kChunkBytes so that staging chunk i+1 overlaps the kernels of chunk i)
        auto stage_array = [&](int i) {
            const size_t bytes = i < 5 ? b->read_bytes : b->hap_bytes;
            if (parts) {
                size_t o = offs[i];
                for (size_t s = 0; s < parts->src[i].size(); ++s) {
                    const size_t n = i < 5 ? parts->read_bytes[s] : parts->hap_bytes[s];
                    if (n) memcpy(A.host + o, parts->src[i][s], n);
                    o += n;
                }
            } else if (bytes) {
                memcpy(A.host + offs[i], src[i], bytes);
            }
        };
        if (st == PHMM_OK) {
            // A chunk of a large call is five arrays of megabytes: the quality tracks go through helper threads while this one
            // copies the bases and the haplotypes (1 536 mixed regions: 0.7-1.1 ms of staging per chunk on the calling thread,
            // a quarter of the call).  Small calls stay on the calling thread -- a thread costs more than their copies.
            if (b->read_bytes >= (1u << 20)) {
                std::thread helpers[4];
                int started = 0;
                try {
                    for (int i = 1; i <= 4; ++i) {
                        helpers[started] = std::thread(stage_array, i);
                        ++started;
                    }
                } catch (const std::system_error &) {  // no thread to be had: the rest on this one
                }
                stage_array(0);
                stage_array(5);
                for (int i = started + 1; i <= 4; ++i) stage_array(i);
                for (int i = 0; i < started; ++i) helpers[i].join();
            } else {
                for (int i = 0; i < 6; ++i) stage_array(i);
            }
        }
        const size_t in_bytes = align_up(A.used, 256);
        b->out_arena_off = in_bytes;
        A.used = in_bytes + 256 + b->n_out * 8;
        if (st != PHMM_OK || A.used > A.cap) {  // cannot happen: batch_create_impl reserved for exactly this layout
            h->err = "phmm_compute: internal error, arena too small";
            std::string keep = h->err;
            phmm_batch_destroy(b);
            h->err = keep;
            return PHMM_ERR_HIP;
        }
        t_stage = now();
        memset(A.host + in_bytes, 0, 256);  // status word
        b->d_status = (uint32_t *)(A.dev + in_bytes);
        double *d_out = (double *)(A.dev + in_bytes + 256);
        // A small one-shot call gets no D2H copy at all: its kernels store the few results straight into the pinned
        // mirror (8 bytes per pair over PCIe), and finish_compute applies the reference's `<= 0` check on the host.
        // (a combined flush of phmm_submit too: nothing on this path is a copy that could get in another flush's way)
        zero_copy = zero_copy_allowed(h) && b->tight_out && b->n_out * 8 <= kZeroCopyOutBytes;
        if (zero_copy) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, A.host + in_bytes + 256, 0) == hipSuccess && dp)
                d_out = (double *)dp;
            else
                zero_copy = false;
        }
        // ... and its inputs are fetched from the mirror by a kernel instead of the copy engine (no cross-engine
        // dependency in front of the first launch; phmm_cigar_kernels.hip)
        void *host_dp = nullptr;
        if (zero_copy && in_bytes + 256 <= kStageInBytes && hipHostGetDevicePointer(&host_dp, A.host, 0) == hipSuccess && host_dp) {
            canary_staged(h, A, in_bytes);
            if (!hip_ok(h, launch_stage_in(host_dp, A.dev, in_bytes + 256, h->S()), "phmm_stage_in_kernel")) st = PHMM_ERR_HIP;
        } else if (!hip_ok(h, hipMemcpyAsync(A.dev, A.host, in_bytes + 256, hipMemcpyHostToDevice, h->S()), "H2D batch")) {
            st = PHMM_ERR_HIP;
        }
        // (slots the kernels never write -- gaps the caller left in out_off -- are never copied back either)
        if (st == PHMM_OK) st = phmm_batch_bind_device(b, d[0], d[1], d[2], d[3], d[4], d[5], d_out);
        t_h2d = now();
    }
    if (st == PHMM_OK) st = phmm_batch_launch(b, nullptr);
    t_launch = now();
    const bool eager = eager_d2h(h);  // otherwise finish_compute fetches the results
    if (st == PHMM_OK && eager && !zero_copy &&
        !hip_ok(h, hipMemcpyAsync(A.host + b->out_arena_off, A.dev + b->out_arena_off, 256 + b->n_out * 8,
                                  hipMemcpyDeviceToHost, h->S()),
                "D2H results"))
        st = PHMM_ERR_HIP;
    if (st != PHMM_OK) {
        std::string keep = h->err;
        (void)hipStreamSynchronize(h->S());
        phmm_batch_destroy(b);
        h->err = keep;
        return st;
    }
    if (trace)
        fprintf(stderr, "  enqueue %u regions: plan %.0f us, stage %.0f us, H2D enqueue %.0f us, launch %.0f us, D2H enqueue %.0f us\n",
                n_regions, t_plan - t_0, t_stage - t_plan, t_h2d - t_stage, t_launch - t_h2d, now() - t_launch);
    pending->b = b;
    pending->slot = h->slot;
    pending->out = out;
    pending->parts = parts;
    pending->stream = h->S();
    pending->d2h_pending = !eager && !zero_copy;
    pending->zero_copy = zero_copy;
    return PHMM_OK;
}

// Next chunk after `c` of the regions [.., n_regions) (start with c.g1 == first region, c.started == false): grows while
// every per-base array stays below the direct-copy limit; `whole` takes everything that is left in one chunk.
bool next_chunk(ChunkView &c, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                const uint32_t *read_off, const uint32_t *hap_off, const uint64_t *out_off, bool whole) {
    const uint32_t g0 = c.g1;
    if (g0 >= n_regions) return false;
    uint32_t g1 = g0 + 1;
    const size_t base_r = read_off[region_read_off[g0]];
    // the first chunks are short so that the GPU starts early (0.5, 0.5, 1, 2, 4, 4 ... MB per array); staging and the
    // H2D copy of the following ones hide behind its kernels
    c.index = c.started ? c.index + 1 : 0;
    c.started = true;
    // f64: 0.5 MB per array four times, then 1, 1, 2, 2, 4, 4 ... (mid-size batches want many small chunks, large ones
    // large launches).  f32-first handles: 1, 2, 4, 4 ... -- the f32 sweep only exists as the chained kernel, which needs
    // a few hundred regions per launch.
    // Mixed batches (round 4): 4 MB per array, then 8, 16, 32 ... -- every chunk of a long-tailed mix is a launch per range of K
    // with a tail of its own, and since planning and staging take 0.85 instead of 2 ms per 4 MB the device no longer waits for
    // the host: 1 536 mixed regions 24.0 ms in seven chunks of 4 MB, 20.4 in three of 4 / 8 / 16 (resident: 15.8).
    // Round 5, the cap swept on one box (tools/hostpath_ragged_sweep.py, best of six calls, twice): 4 / 8 / 16: 18.75-18.85 ms;
    // 4 / 8 / 8 / 7: 18.1-18.3; cap 6: 19.3-19.4; cap 12: 18.7-18.8; first chunk 3 / 6 MB: 18.5 / 20.1; a first chunk of 1 or 2 MB and
    // 8 MB ones behind it (the device starts 0.5 ms earlier, one more set of launches): 18.1-18.3 -- so the cap is 8 MB.
    const uint32_t step = c.f32_first ? c.index + 1 : c.mixed ? c.index : (c.index < 4 ? 0 : (c.index - 2) / 2);
    const size_t max_bytes = c.mixed && !c.f32_first ? kMixedChunkBytes : kChunkBytes;
    const size_t limit = std::min(max_bytes, (c.f32_first ? (1u << 20) / 2 : c.mixed ? kMixedFirstChunkBytes : kFirstChunkBytes) << std::min<uint32_t>(step, 16));
    if (whole) {
        g1 = n_regions;
    } else if (!c.mixed) {
        while (g1 < n_regions && (size_t)read_off[region_read_off[g1 + 1]] - base_r <= limit) ++g1;
    } else {
        // A mixed batch falls into many shape classes, and a chunk of a few dozen regions is too small for the chained
        // kernel: every class then gets a small per-read launch of its own (1 536 mixed regions: the first six chunks
        // were 10-20 launches of ~5 us each for 184 regions).  Such a chunk keeps growing until it has enough wave-sweeps
        // to chain (or reaches the largest chunk size).
        auto units = [&](uint32_t g) {
            return (uint64_t)(region_read_off[g + 1] - region_read_off[g]) * ((region_hap_off[g + 1] - region_hap_off[g] + 3) / 4);
        };
        uint64_t u = units(g0);
        while (g1 < n_regions) {
            const size_t bytes = (size_t)read_off[region_read_off[g1 + 1]] - base_r;
            if (bytes > max_bytes || (bytes > limit && u >= 8ull * 2 * kNumSimd * 4)) break;
            u += units(g1);
            ++g1;
        }
        // ... and a remainder too small to chain on its own rides with this chunk instead of following as a per-read launch
        if (g1 < n_regions) {
            uint64_t rest = 0;
            for (uint32_t g = g1; g < n_regions && rest < 8ull * 2 * kNumSimd * 4; ++g) rest += units(g);
            if (rest < 8ull * 2 * kNumSimd * 4) g1 = n_regions;
        }
    }
    c.g0 = g0;
    c.g1 = g1;
    c.r0 = region_read_off[g0];
    c.r1 = region_read_off[g1];
    c.h0 = region_hap_off[g0];
    c.h1 = region_hap_off[g1];
    c.read_byte0 = read_off[c.r0];
    c.hap_byte0 = hap_off[c.h0];
    c.rro.resize(g1 - g0 + 1);
    c.rho.resize(g1 - g0 + 1);
    c.oo.resize(g1 - g0 + 1);
    for (uint32_t g = g0; g <= g1; ++g) {
        c.rro[g - g0] = region_read_off[g] - c.r0;
        c.rho[g - g0] = region_hap_off[g] - c.h0;
        c.oo[g - g0] = out_off[g] - out_off[g0];
    }
    c.ro.resize(c.r1 - c.r0 + 1);
    for (uint32_t r = c.r0; r <= c.r1; ++r) c.ro[r - c.r0] = read_off[r] - read_off[c.r0];
    c.ho.resize(c.h1 - c.h0 + 1);
    for (uint32_t a = c.h0; a <= c.h1; ++a) c.ho[a - c.h0] = hap_off[a] - hap_off[c.h0];
    return true;
}

// Wait for a pending batch, hand the results to the caller, release the batch.
int finish_compute(phmm_handle *h, PendingCompute *p) {
    if (!p->b) return PHMM_OK;
    int st = PHMM_OK;
    phmm_batch *b = p->b;
    Arena &A = h->arenas[p->slot];
    hipStream_t S = p->stream ? p->stream : h->streams[p->slot];
    const size_t res_bytes = 256 + b->n_out * 8;
    auto fetch = [&]() {  // [status | out] -> pinned mirror
        return hip_ok(h, hipMemcpyAsync(A.host + b->out_arena_off, A.dev + b->out_arena_off, res_bytes, hipMemcpyDeviceToHost, S),
                      "D2H results") &&
               hip_ok(h, wait_stream(h, S), "sync(D2H)");
    };
    if (!hip_ok(h, wait_stream(h, S), "sync") || (p->d2h_pending && !fetch())) {  // kernels are done: fetch now
        st = PHMM_ERR_HIP;
    } else {
        const char *hs = A.host + b->out_arena_off;
        const double *v = (const double *)(hs + 256);
        uint32_t bits = *(const uint32_t *)hs;
        auto each_result = [&](auto &&f) {  // every slot a kernel wrote
            if (b->tight_out)
                for (uint64_t i = 0; i < b->n_out; ++i) f(v[i]);
            else
                for (const auto &e : b->out_extents)
                    for (uint64_t i = 0; i < e.second; ++i) f(v[e.first + i]);
        };
        if (p->zero_copy) {  // the status word stayed on the device: every slot was written once, look at the values
            bits = 0;
            each_result([&](double x) { bits |= status_bits(x); });
        }
        if ((bits & STATUS_RESCUE) && !h->sw.no_rescue) {
            // some pair came out below kRescueBelow: redo those in the reference's operation order, fetch again
            uint32_t nb = 0;
            h->stat_rescue_passes += 1;
            if (!ensure_arena_rescue(h, A, b->max_h, &nb) || launch_rescue_pass(b, A.rescue, nb, true, S) != PHMM_OK ||
                !hip_ok(h, hipStreamSynchronize(S), "sync(rescue)") || (!p->zero_copy && !fetch())) {
                st = PHMM_ERR_HIP;
            } else {
                // (from the values on every path: phmm_rescue only ORs into the device word, so a bit a fast kernel raised
                // for a pair the exact pass has since replaced would survive there)
                bits = 0;
                each_result([&](double x) { bits |= status_bits(x); });
            }
        }
        if (st == PHMM_OK) {
            const char *src = hs + 256;
            if (p->parts) {
                // part s owns the regions [first_region[s], first_region[s+1]) of the combined batch
                for (size_t s = 0; s < p->parts->out.size(); ++s) {
                    if (b->tight_out) {
                        if (p->parts->n_out[s]) memcpy(p->parts->out[s], src, p->parts->n_out[s] * 8);
                    } else {
                        const uint32_t g0 = p->parts->first_region[s], g1 = p->parts->first_region[s + 1];
                        for (uint32_t g = g0; g < g1; ++g) {  // the part's out_off starts at 0, like every offset array
                            const auto &e = b->out_extents[g];
                            if (e.second) memcpy(p->parts->out[s] + (e.first - b->out_extents[g0].first), v + e.first, e.second * 8);
                        }
                    }
                    src += p->parts->n_out[s] * 8;
                }
            } else if (b->tight_out) {
                if (b->n_out) memcpy(p->out, src, b->n_out * 8);
            } else {
                for (const auto &e : b->out_extents)
                    if (e.second) memcpy(p->out + e.first, v + e.first, e.second * 8);
            }
            if (bits & STATUS_POSITIVE) {
                h->err = "PairHmm Log Probability cannot be greater than 0.0";  // pair_hmm.rs:478-481
                st = PHMM_ERR_POSITIVE_RESULT;
            }
            // (PHMM_MIRROR_CANARY: the results are with the caller -- nothing may store into this call's block from here on)
            if (p->zero_copy && !canary_after_call(h, A, b->out_arena_off, res_bytes) && st == PHMM_OK) st = PHMM_ERR_INTERNAL;
        }
    }
    std::string keep = h->err;
    phmm_batch_destroy(b);
    h->err = keep;
    p->b = nullptr;
    return st;
}


phmm_batch *batch_create_in_arena(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                                  const uint32_t *read_off, const uint32_t *hap_off, const uint64_t *out_off, size_t extra_arena_bytes) {
    return batch_create_impl(h, n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, true, extra_arena_bytes);
}
BatchView batch_view(const phmm_batch *b) {
    BatchView v{};
    v.n_regions = b->n_regions;
    v.n_reads = b->n_reads;
    v.n_haps = b->n_haps;
    v.max_h = b->max_h;
    v.n_out = b->n_out;
    v.read_bytes = b->read_bytes;
    v.hap_bytes = b->hap_bytes;
    v.tight_out = b->tight_out;
    v.d_read_region = b->d_read_region;
    v.d_region_read_off = b->d_region_read_off;
    v.d_region_hap_off = b->d_region_hap_off;
    v.d_read_off = b->d_read_off;
    v.d_hap_off = b->d_hap_off;
    v.d_out_off = b->d_out_off;
    return v;
}
void batch_set_status(phmm_batch *b, uint32_t *d_status) { b->d_status = d_status; }
bool batch_set_inline_rescue(phmm_handle *h, phmm_batch *b) {
    if (!b->n_reads || h->sw.no_rescue) return true;
    if (!ensure_arena_rescue(h, *b->arena, b->max_h, &b->rescue_blocks)) return false;
    b->rescue_scratch = b->arena->rescue;
    return true;
}
void batch_copy_out(const phmm_batch *b, const double *src, double *out) {
    if (b->tight_out) {
        if (b->n_out) memcpy(out, src, b->n_out * 8);
    } else {  // gaps the caller left in out_off stay untouched
        for (const auto &e : b->out_extents)
            if (e.second) memcpy(out + e.first, src + e.first, e.second * 8);
    }
}
size_t one_shot_bytes() { return kOneShotBytes; }
size_t stage_in_bytes() { return kStageInBytes; }
size_t zero_copy_out_bytes() { return kZeroCopyOutBytes; }

}  // namespace phmm_host

extern "C" {

}  // extern "C"

namespace phmm_host {

// phmm_compute on the regions [g_begin, g_end) of the caller's (already validated) arrays; results go where
// phmm_compute on the whole batch would put them.  phmm_compute is the range [0, n_regions); phmm_compute_multi hands
// every engine a contiguous range -- no gather, every payload byte is copied once, into the pinned mirror.
int compute_range(phmm_handle *h, uint32_t g_begin, uint32_t g_end, const uint32_t *region_read_off,
                  const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                  const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off,
                  const uint8_t *hap_bases, const uint64_t *out_off, double *out) {
    const bool trace = h->sw.trace != 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    DeviceGuard dg(h->device);
    const uint32_t n_regions = g_end - g_begin;
    const size_t range_bytes = (size_t)read_off[region_read_off[g_end]] - read_off[region_read_off[g_begin]];
    // ---- small / medium batch: one shot ------------------------------------------------------------
    const bool f32_first = (h->flags & PHMM_FLAG_F32_FIRST) != 0;
    if (n_regions < 8 || range_bytes <= (f32_first ? (size_t)(8u << 20) : kOneShotBytes) || h->sw.no_pipeline) {
        h->slot = 0;
        PendingCompute p;
        int st;
        if (g_begin == 0) {  // offsets already start at 0
            st = enqueue_compute(h, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, gcp,
                                 hap_off, hap_bases, out_off, out, &p);
        } else {
            ChunkView c;
            c.g1 = g_begin;
            (void)next_chunk(c, g_end, region_read_off, region_hap_off, read_off, hap_off, out_off, true);
            const size_t bo = c.read_byte0, co = c.hap_byte0;
            st = enqueue_compute(h, n_regions, c.rro.data(), c.rho.data(), c.ro.data(), read_bases + bo, base_q + bo, ins_q + bo,
                                 del_q + bo, gcp + bo, c.ho.data(), hap_bases + co, c.oo.data(), out + out_off[g_begin], &p);
        }
        const double t1 = now();
        if (st == PHMM_OK) st = finish_compute(h, &p);
        if (trace)
            fprintf(stderr, "phmm_compute: plan+stage+enqueue %.1f us, wait+copy-out %.1f us\n", t1 - t0, now() - t1);
        return st;
    }
    // ---- large batch: chunks of regions rotate through kSlots (arena, stream) pairs, so the host staging
    //      and the H2D copy of chunk i+1 overlap the kernels of chunk i.  Results are identical: every
    //      region is independent, the chunk only rebases the offsets. -----------------------------------
    PendingCompute pend[kSlots];
    struct Drain {  // an exception on the way (host allocation) must not leave chunks in flight
        phmm_handle *h;
        PendingCompute *pend;
        ~Drain() {
            for (int i = 0; i < kSlots; ++i)
                if (pend[i].b) {
                    (void)hipStreamSynchronize(h->streams[pend[i].slot]);
                    phmm_batch_destroy(pend[i].b);
                    pend[i].b = nullptr;
                }
            h->slot = 0;
            h->defer_d2h = false;
        }
    } drain{h, pend};
    int st = PHMM_OK;
    ChunkView c;
    c.f32_first = f32_first;
    c.g1 = g_begin;
    {   // same read and haplotype counts and the same first haplotype length everywhere?
        const uint32_t nr0 = region_read_off[g_begin + 1] - region_read_off[g_begin];
        const uint32_t nh0 = region_hap_off[g_begin + 1] - region_hap_off[g_begin];
        const uint32_t hl0 = nh0 ? hap_off[region_hap_off[g_begin] + 1] - hap_off[region_hap_off[g_begin]] : 0;
        for (uint32_t g = g_begin + 1; g < g_end && !c.mixed; ++g)
            c.mixed = region_read_off[g + 1] - region_read_off[g] != nr0 || region_hap_off[g + 1] - region_hap_off[g] != nh0 ||
                      (nh0 && hap_off[region_hap_off[g] + 1] - hap_off[region_hap_off[g]] != hl0);
    }
    int n_chunks = 0;
    h->defer_d2h = true;
    while (st == PHMM_OK && next_chunk(c, g_end, region_read_off, region_hap_off, read_off, hap_off, out_off)) {
        const int slot = n_chunks % kSlots;
        st = finish_compute(h, &pend[slot]);  // the slot's previous chunk must be out of its arena
        if (st != PHMM_OK) break;
        h->slot = slot;
        const size_t bo = c.read_byte0, co = c.hap_byte0;
        st = enqueue_compute(h, c.g1 - c.g0, c.rro.data(), c.rho.data(), c.ro.data(), read_bases + bo, base_q + bo, ins_q + bo,
                             del_q + bo, gcp + bo, c.ho.data(), hap_bases + co, c.oo.data(), out + out_off[c.g0], &pend[slot]);
        ++n_chunks;
    }
    for (int i = 0; i < kSlots; ++i) {  // drain in submission order
        const int slot = (n_chunks + i) % kSlots;
        const int s2 = finish_compute(h, &pend[slot]);
        if (st == PHMM_OK) st = s2;
    }
    h->slot = 0;
    h->defer_d2h = false;
    if (trace) fprintf(stderr, "phmm_compute: %d chunks pipelined over %d slots, total %.1f us\n", n_chunks, kSlots, now() - t0);
    return st;
}

// The regions `list` (any order, any subset) of the caller's arrays on one engine: chunks of the list are staged
// straight from the caller's arrays (a scatter-gather `Parts` entry per region) and rotate through the engine's slots
// like the chunks of compute_range.  What phmm_compute_multi uses when a heavy-tailed set makes contiguous ranges
// unbalanced and regions are dealt out one by one.
int compute_list(phmm_handle *h, const uint32_t *list, uint32_t n_list, const uint32_t *region_read_off,
                 const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                 const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off,
                 const uint8_t *hap_bases, const uint64_t *out_off, double *out) {
    DeviceGuard dg(h->device);
    struct Slot {
        PendingCompute pend;
        Parts parts;
        std::vector<uint32_t> rro, rho, ro, ho;
        std::vector<uint64_t> oo;
    } slots[kSlots];
    struct Drain {
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
        }
    } drain{h, slots};
    const uint8_t *src[6] = {read_bases, base_q, ins_q, del_q, gcp, hap_bases};
    int st = PHMM_OK, n_chunks = 0;
    h->defer_d2h = true;
    for (uint32_t i0 = 0; i0 < n_list && st == PHMM_OK;) {
        Slot &S = slots[n_chunks % kSlots];
        st = finish_compute(h, &S.pend);  // the slot's previous chunk must be out of its arena (and of S.parts)
        if (st != PHMM_OK) break;
        const size_t limit = std::min(kChunkBytes, kFirstChunkBytes << std::min(n_chunks / 2, 16));
        S.rro.assign(1, 0);
        S.rho.assign(1, 0);
        S.ro.assign(1, 0);
        S.ho.assign(1, 0);
        S.oo.assign(1, 0);
        for (int k = 0; k < 6; ++k) S.parts.src[k].clear();
        S.parts.read_bytes.clear();
        S.parts.hap_bytes.clear();
        S.parts.out.clear();
        S.parts.n_out.clear();
        S.parts.first_region.assign(1, 0);
        uint32_t i1 = i0;
        size_t bytes = 0;
        while (i1 < n_list && S.parts.out.size() < 4096) {
            const uint32_t g = list[i1];
            const uint32_t r0 = region_read_off[g], r1 = region_read_off[g + 1], a0 = region_hap_off[g], a1 = region_hap_off[g + 1];
            const size_t rb = (size_t)read_off[r1] - read_off[r0], hb = (size_t)hap_off[a1] - hap_off[a0];
            if (i1 > i0 && bytes + rb > limit) break;
            bytes += rb;
            for (uint32_t r = r0; r < r1; ++r) S.ro.push_back(S.ro.back() + (read_off[r + 1] - read_off[r]));
            for (uint32_t a = a0; a < a1; ++a) S.ho.push_back(S.ho.back() + (hap_off[a + 1] - hap_off[a]));
            S.rro.push_back(S.rro.back() + (r1 - r0));
            S.rho.push_back(S.rho.back() + (a1 - a0));
            S.oo.push_back(S.oo.back() + (out_off[g + 1] - out_off[g]));
            for (int k = 0; k < 5; ++k) S.parts.src[k].push_back(src[k] + read_off[r0]);
            S.parts.src[5].push_back(hap_bases + hap_off[a0]);
            S.parts.read_bytes.push_back(rb);
            S.parts.hap_bytes.push_back(hb);
            S.parts.out.push_back(out + out_off[g]);
            S.parts.n_out.push_back(out_off[g + 1] - out_off[g]);
            S.parts.first_region.push_back(S.parts.first_region.back() + 1);
            ++i1;
        }
        h->slot = n_chunks % kSlots;
        st = enqueue_compute(h, i1 - i0, S.rro.data(), S.rho.data(), S.ro.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                             S.ho.data(), nullptr, S.oo.data(), nullptr, &S.pend, &S.parts);
        ++n_chunks;
        i0 = i1;
    }
    for (int i = 0; i < kSlots; ++i) {
        const int s2 = finish_compute(h, &slots[(n_chunks + i) % kSlots].pend);
        if (st == PHMM_OK) st = s2;
    }
    h->slot = 0;
    h->defer_d2h = false;
    return st;
}

}  // namespace phmm_host

extern "C" {

int phmm_compute(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                 const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q,
                 const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off, const uint8_t *hap_bases,
                 const uint64_t *out_off, double *out) {
    if (!h) return PHMM_ERR_INVALID_ARG;
    PHMM_GUARD_BEGIN
    phmm_host::latch_slot0(h);
    // the whole batch is checked before anything is indexed: the chunked path walks the caller's arrays
    h->err_code = PHMM_OK;
    if (tl_err_h == h) tl_err_h = nullptr;
    if (const char *bad = validate_offsets(n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, nullptr)) {
        h->err = bad;
        return h->err_code = PHMM_ERR_INVALID_ARG;
    }
    const uint32_t n_reads = region_read_off[n_regions];
    if ((read_off[n_reads] && (!read_bases || !base_q || !ins_q || !del_q || !gcp)) ||
        (hap_off[region_hap_off[n_regions]] && !hap_bases) || (out_off[n_regions] && !out)) {
        h->err = "phmm_compute: null pointer";
        return h->err_code = PHMM_ERR_INVALID_ARG;
    }
    // (one of many private handles on the device: opt-in, PHMM_ROUTE_SHARED, a one-shot call goes through the device's shared handle.
    // The PairHMM alone through the region server -- built in round 6 -- gave a wrong likelihood once in 150 000 calls of the ragged
    // mix under TB_VERIFY and was taken out again; the whole region call, whose waves go on to the alignment, soaks clean.)
    if (n_regions < 8 || (size_t)read_off[n_reads] <= kOneShotBytes)
        if (phmm_handle *via = route_shared(h)) {
            uint64_t ticket = 0;
            int st = phmm_submit(via, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, gcp, hap_off, hap_bases,
                                 out_off, out, &ticket);
            if (st == PHMM_OK) st = phmm_wait(via, ticket);
            if (st != PHMM_OK) {
                h->err = phmm_last_error(via);
                h->err_code = st;
            }
            return st;
        }
    return compute_range(h, 0, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, gcp,
                         hap_off, hap_bases, out_off, out);
    PHMM_GUARD_END(h, "phmm_compute", PHMM_FAIL_CODE)
}

namespace {

struct PendingEngine {
    phmm_batch *b = nullptr;
    int slot = 0;
    hipStream_t stream = nullptr;  // the stream the batch was enqueued on
    bool zero_copy = false;        // the kernels stored the results into the pinned mirror
    double *out = nullptr;
    uint8_t *keep = nullptr;
    size_t res_off = 0, keep_bytes = 0, res_bytes = 0;
    uint32_t n_reads = 0;
    bool d2h_pending = false;
};

// Stage one batch of the engine-level call in the current slot's arena and enqueue H2D, pre-step, PairHMM, post-step
// and D2H on its stream.  No sync.
int engine_enqueue(phmm_handle *h, const phmm_engine_config *cfg, uint32_t n_regions, const uint32_t *region_read_off,
                   const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                   const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *mapq, const uint32_t *hap_off,
                   const uint8_t *hap_bases, const int32_t *region_ref_hap, const uint64_t *out_off, double *out, uint8_t *keep,
                   PendingEngine *pending) {
    const uint32_t n_reads = region_read_off[n_regions];
    const size_t rbytes = align_up((size_t)read_off[n_reads], 256);
    // originals (4 x read bytes + mapq + ref index) and device-only copies (4 x read bytes, thresholds, keep)
    const size_t extra = 4 * rbytes + 4 * rbytes + align_up((size_t)n_reads, 256) * 2 + align_up((size_t)n_reads * 8, 256) +
                         align_up((size_t)n_regions * 4, 256) + 18 * 256;
    phmm_batch *b = batch_create_impl(h, n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, true, extra);
    if (!b) return h->err_code ? h->err_code : PHMM_ERR_INVALID_ARG;
    int st = PHMM_OK;
    if ((b->read_bytes && (!read_bases || !base_q)) || (n_reads && (!mapq || !keep)) || (b->hap_bytes && !hap_bases) ||
        (b->n_out && !out)) {
        h->err = "phmm_engine_compute: null pointer";
        st = PHMM_ERR_INVALID_ARG;
    }
    Arena &A = h->A();
    if (st == PHMM_OK) {
        bool fits = true;
        auto place = [&](const void *src, size_t bytes) -> char * {  // into the mirror (travels in the one H2D copy)
            const size_t off = align_up(A.used, 256);
            if (off + bytes > A.cap) {
                fits = false;
                return A.dev;
            }
            A.used = off + bytes;
            if (src && bytes) memcpy(A.host + off, src, bytes);
            return A.dev + off;
        };
        const uint8_t *d_bases = (const uint8_t *)place(read_bases, b->read_bytes);
        const uint8_t *d_q0 = (const uint8_t *)place(base_q, b->read_bytes);
        const uint8_t *d_i0 = ins_q ? (const uint8_t *)place(ins_q, b->read_bytes) : nullptr;
        const uint8_t *d_d0 = del_q ? (const uint8_t *)place(del_q, b->read_bytes) : nullptr;
        const uint8_t *d_mapq = (const uint8_t *)place(mapq, n_reads);
        const uint8_t *d_haps = (const uint8_t *)place(hap_bases, b->hap_bytes);
        const int32_t *d_ref = region_ref_hap ? (const int32_t *)place(region_ref_hap, (size_t)n_regions * 4) : nullptr;
        // (a zeroed status block that travels with the inputs: what the kernels of a small call flag, see below)
        uint32_t *d_status_in = (uint32_t *)place(nullptr, 256);
        if (fits) memset(A.host + ((char *)d_status_in - A.dev), 0, 256);
        const size_t in_bytes = align_up(A.used, 256);
        // device-only
        uint8_t *d_q = (uint8_t *)place(nullptr, b->read_bytes), *d_i = (uint8_t *)place(nullptr, b->read_bytes),
                *d_d = (uint8_t *)place(nullptr, b->read_bytes), *d_g = (uint8_t *)place(nullptr, b->read_bytes);
        double *d_thr = (double *)place(nullptr, (size_t)n_reads * 8);
        // results: [status (256 B) | keep | out] contiguous, one D2H
        const size_t res_off = align_up(A.used, 256);
        const size_t keep_bytes = align_up((size_t)n_reads, 256);
        A.used = res_off + 256 + keep_bytes + b->n_out * 8;
        if (!fits || A.used > A.cap) {  // cannot happen: `extra` above reserves for exactly this layout
            h->err = "phmm_engine_compute: internal error, arena too small";
            std::string keep_err = h->err;
            phmm_batch_destroy(b);
            h->err = keep_err;
            return PHMM_ERR_HIP;
        }
        b->d_status = (uint32_t *)(A.dev + res_off);
        uint8_t *d_keep = (uint8_t *)(A.dev + res_off + 256);
        double *d_out = (double *)(A.dev + res_off + 256 + keep_bytes);
        const size_t res_bytes = 256 + keep_bytes + b->n_out * 8;
        // A small one-shot call (a region per call, the reference's pattern) does without the copy engine, like
        // phmm_compute: a kernel fetches the inputs from the pinned mirror, and the post-step stores keep flags and
        // normalised likelihoods -- and hands on the status word -- straight into it.
        char *mirror = nullptr;
        if (zero_copy_allowed(h) && b->tight_out && n_reads && in_bytes <= kStageInBytes && res_bytes <= kZeroCopyOutBytes) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, A.host, 0) == hipSuccess && dp) mirror = (char *)dp;
        }
        bool ok;
        if (mirror) {
            b->d_status = d_status_in;
            canary_staged(h, A, in_bytes);
            ok = hip_ok(h, launch_stage_in(mirror, A.dev, in_bytes, h->S()), "phmm_stage_in_kernel");
        } else {
            ok = hip_ok(h, hipMemcpyAsync(A.dev, A.host, in_bytes, hipMemcpyHostToDevice, h->S()), "H2D batch") &&
                 hip_ok(h, hipMemsetAsync(b->d_status, 0, 256, h->S()), "memset status");
        }
        // the post-step consumes the likelihoods on the device, so the exact pass below kRescueBelow rides in-stream
        // between the forward kernels and the post-step (phmm_batch_launch); nothing of this slot is in flight now
        if (ok && n_reads && !h->sw.no_rescue) {
            ok = ensure_arena_rescue(h, A, b->max_h, &b->rescue_blocks);
            if (ok) b->rescue_scratch = A.rescue;
        }
        uint32_t max_r = 0;
        for (uint32_t r = 0; r < n_reads; ++r) max_r = std::max(max_r, read_off[r + 1] - read_off[r]);
        PrepParams pp{};
        pp.n_reads = n_reads;
        pp.read_off = b->d_read_off;
        pp.read_bases = d_bases;
        pp.base_q = d_q0;
        pp.ins_q = d_i0;
        pp.del_q = d_d0;
        pp.mapq = d_mapq;
        pp.pcr_cache = cfg->pcr_error_model ? h->d_pcr_cache + 128 * cfg->pcr_error_model : nullptr;
        pp.out_q = d_q;
        pp.out_ins = d_i;
        pp.out_del = d_d;
        pp.out_gcp = d_g;
        pp.threshold = d_thr;
        pp.lds_rows = (uint32_t)align_up((size_t)max_r + 1, 8);
        pp.waves_per_read = n_reads <= 2048 ? std::max<uint32_t>(1, (max_r + 63) / 64) : 1;  // few reads: a wave per 64 positions
        pp.default_indel_qual = 45;  // ReadUtils::DEFAULT_INSERTION_DELETION_QUAL (read_utils.rs:23)
        pp.constant_gcp = cfg->constant_gcp;
        pp.base_quality_score_threshold = cfg->base_quality_score_threshold;
        pp.disable_cap_to_mapq = cfg->disable_cap_read_qualities_to_mapq;
        pp.dynamic_disqualification = cfg->dynamic_read_disqualification;
        pp.read_disqualification_scale = cfg->read_disqualification_scale;
        pp.expected_error_rate_per_base = cfg->expected_error_rate_per_base;
        if (ok && (size_t)pp.lds_rows * 17 * 4 > kLdsBytesPerCU) {
            h->err = "phmm_engine_compute: read too long for the pre-step kernel";
            ok = false;
            st = PHMM_ERR_INVALID_ARG;
        }
        if (ok) ok = hip_ok(h, launch_prep(pp, h->S()), "phmm_prep_reads");
        if (ok) ok = phmm_batch_bind_device(b, d_bases, d_q, d_i, d_d, d_g, d_haps, d_out) == PHMM_OK;
        if (ok) ok = phmm_batch_launch(b, nullptr) == PHMM_OK;
        PostParams po{};
        po.n_reads = n_reads;
        po.read_region = b->d_read_region;
        po.region_read_off = b->d_region_read_off;
        po.region_hap_off = b->d_region_hap_off;
        po.out_off = b->d_out_off;
        po.region_ref_hap = d_ref;
        po.out = d_out;
        po.out_final = mirror ? (double *)(mirror + res_off + 256 + keep_bytes) : nullptr;
        po.threshold = d_thr;
        po.keep = mirror ? (uint8_t *)(mirror + res_off + 256) : d_keep;
        po.status_in = mirror ? b->d_status : nullptr;
        po.status_out = mirror ? (uint32_t *)(mirror + res_off) : nullptr;
        po.max_likelihood_difference_cap = cfg->log10_global_read_mismapping_rate;
        po.symmetric = cfg->symmetrically_normalize_alleles_to_reference;
        if (ok) ok = hip_ok(h, launch_post(po, h->S()), "phmm_post_reads");
        const bool eager = eager_d2h(h);  // otherwise engine_finish fetches the results
        if (ok && eager && !mirror)
            ok = hip_ok(h, hipMemcpyAsync(A.host + res_off, A.dev + res_off, res_bytes, hipMemcpyDeviceToHost, h->S()),
                        "D2H results");
        if (ok) {
            pending->res_bytes = res_bytes;
            pending->d2h_pending = !eager && !mirror;
            pending->b = b;
            pending->slot = h->slot;
            pending->out = out;
            pending->keep = keep;
            pending->stream = h->S();
            pending->zero_copy = mirror != nullptr;
            pending->res_off = res_off;
            pending->keep_bytes = keep_bytes;
            pending->n_reads = n_reads;
            return PHMM_OK;
        }
        if (st == PHMM_OK) st = PHMM_ERR_HIP;
    }
    std::string keep_err = h->err;
    (void)hipStreamSynchronize(h->S());
    phmm_batch_destroy(b);
    h->err = keep_err;
    return st;
}

// Wait for a pending engine batch, hand keep flags and likelihoods to the caller, release the batch.
int engine_finish(phmm_handle *h, PendingEngine *p) {
    if (!p->b) return PHMM_OK;
    int st = PHMM_OK;
    phmm_batch *b = p->b;
    Arena &A = h->arenas[p->slot];
    hipStream_t S = p->stream ? p->stream : h->streams[p->slot];
    if (!hip_ok(h, wait_stream(h, S), "sync") ||
        (p->d2h_pending &&
         (!hip_ok(h, hipMemcpyAsync(A.host + p->res_off, A.dev + p->res_off, p->res_bytes, hipMemcpyDeviceToHost, S),
                  "D2H results") ||
          !hip_ok(h, wait_stream(h, S), "sync(D2H)")))) {
        st = PHMM_ERR_HIP;
    } else {
        const char *hs = A.host + p->res_off;
        if (p->n_reads) memcpy(p->keep, hs + 256, p->n_reads);
        const double *v = (const double *)(hs + 256 + p->keep_bytes);
        if (b->tight_out) {
            if (b->n_out) memcpy(p->out, v, b->n_out * 8);
        } else {  // gaps the caller left in out_off stay untouched
            for (const auto &e : b->out_extents)
                if (e.second) memcpy(p->out + e.first, v + e.first, e.second * 8);
        }
        if (status_positive(*(const uint32_t *)hs, b->rescue_scratch != nullptr && !h->sw.no_rescue)) {
            h->err = "PairHmm Log Probability cannot be greater than 0.0";  // pair_hmm.rs:478-481
            st = PHMM_ERR_POSITIVE_RESULT;
        }
        if (p->zero_copy && !canary_after_call(h, A, p->res_off, p->res_bytes) && st == PHMM_OK) st = PHMM_ERR_INTERNAL;
    }
    std::string keep_err = h->err;
    phmm_batch_destroy(b);
    h->err = keep_err;
    p->b = nullptr;
    return st;
}

}  // namespace

int phmm_engine_compute(phmm_handle *h, const phmm_engine_config *cfg, uint32_t n_regions,
                        const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                        const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q,
                        const uint8_t *mapq, const uint32_t *hap_off, const uint8_t *hap_bases,
                        const int32_t *region_ref_hap, const uint64_t *out_off, double *out, uint8_t *keep) {
    if (!h || !cfg) return PHMM_ERR_INVALID_ARG;
    PHMM_GUARD_BEGIN
    phmm_host::latch_slot0(h);
    h->err_code = PHMM_OK;
    if (tl_err_h == h) tl_err_h = nullptr;
    if (cfg->pcr_error_model > 3) {
        h->err = "phmm_engine_compute: Unknown PCR Error Model";  // engine.rs:89
        return h->err_code = PHMM_ERR_INVALID_ARG;
    }
    // the whole batch is checked before anything is indexed: the chunked path below walks the caller's arrays
    if (const char *bad = validate_offsets(n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, nullptr)) {
        h->err = bad;
        return h->err_code = PHMM_ERR_INVALID_ARG;
    }
    const uint32_t n_reads = region_read_off[n_regions];
    if ((read_off[n_reads] && (!read_bases || !base_q)) || (n_reads && (!mapq || !keep)) ||
        (hap_off[region_hap_off[n_regions]] && !hap_bases) || (out_off[n_regions] && !out)) {
        h->err = "phmm_engine_compute: null pointer";
        return h->err_code = PHMM_ERR_INVALID_ARG;
    }
    if (n_regions < 8 || (size_t)read_off[n_reads] <= kOneShotBytes)
        if (phmm_handle *via = route_shared(h)) {  // (one of many private handles on the device: phmm_compute has the comment)
            uint64_t ticket = 0;
            int st = phmm_engine_submit(via, cfg, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, mapq, hap_off,
                                        hap_bases, region_ref_hap, out_off, out, keep, &ticket);
            if (st == PHMM_OK) st = phmm_wait(via, ticket);
            if (st != PHMM_OK) {
                h->err = phmm_last_error(via);
                h->err_code = st;
            }
            return st;
        }
    DeviceGuard dg(h->device);
    // ---- small / medium batch: one shot ------------------------------------------------------------
    if (n_regions < 8 || (size_t)read_off[n_reads] <= kOneShotBytes || h->sw.no_pipeline) {
        h->slot = 0;
        PendingEngine p;
        int st = engine_enqueue(h, cfg, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q,
                                mapq, hap_off, hap_bases, region_ref_hap, out_off, out, keep, &p);
        if (st == PHMM_OK) st = engine_finish(h, &p);
        return st;
    }
    // ---- large batch: chunks of regions through the kSlots (arena, stream) pairs, like phmm_compute ----
    PendingEngine pend[kSlots];
    struct Drain {  // an exception on the way (host allocation) must not leave chunks in flight
        phmm_handle *h;
        PendingEngine *pend;
        ~Drain() {
            for (int i = 0; i < kSlots; ++i)
                if (pend[i].b) {
                    (void)hipStreamSynchronize(h->streams[pend[i].slot]);
                    phmm_batch_destroy(pend[i].b);
                    pend[i].b = nullptr;
                }
            h->slot = 0;
            h->defer_d2h = false;
        }
    } drain{h, pend};
    int st = PHMM_OK;
    ChunkView c;
    int n_chunks = 0;
    h->defer_d2h = true;
    while (st == PHMM_OK && next_chunk(c, n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off)) {
        const int slot = n_chunks % kSlots;
        st = engine_finish(h, &pend[slot]);  // the slot's previous chunk must be out of its arena
        if (st != PHMM_OK) break;
        h->slot = slot;
        const size_t bo = c.read_byte0, co = c.hap_byte0;
        st = engine_enqueue(h, cfg, c.g1 - c.g0, c.rro.data(), c.rho.data(), c.ro.data(), read_bases ? read_bases + bo : nullptr,
                            base_q ? base_q + bo : nullptr, ins_q ? ins_q + bo : nullptr, del_q ? del_q + bo : nullptr,
                            mapq ? mapq + c.r0 : nullptr, c.ho.data(), hap_bases ? hap_bases + co : nullptr,
                            region_ref_hap ? region_ref_hap + c.g0 : nullptr, c.oo.data(), out ? out + out_off[c.g0] : nullptr,
                            keep ? keep + c.r0 : nullptr, &pend[slot]);
        ++n_chunks;
    }
    for (int i = 0; i < kSlots; ++i) {  // drain in submission order
        const int slot = (n_chunks + i) % kSlots;
        const int s2 = engine_finish(h, &pend[slot]);
        if (st == PHMM_OK) st = s2;
    }
    h->slot = 0;
    h->defer_d2h = false;
    return st;
    PHMM_GUARD_END(h, "phmm_engine_compute", PHMM_FAIL_CODE)
}

int phmm_set_switch(phmm_handle *h, const char *name, int value) {
    if (!h || !name) return PHMM_ERR_INVALID_ARG;
    Switches &w = h->sw;
    const std::string n(name);
    if (n == "force_L") w.force_L = value > 0 ? value : 0;
    else if (n == "force_chain") w.force_chain = value;
    else if (n == "force_streams") w.force_streams = value > 0 ? value : 0;
    else if (n == "no_pipeline") w.no_pipeline = value != 0;
    else if (n == "no_rescue") w.no_rescue = value != 0;
    else if (n == "trace") w.trace = value != 0;
    else if (n == "sw_lite") {
        w.sw_lite = value;
        h->swork.lite_skip = 0;  // (what earlier calls have taught the handle starts over)
    }
    else if (n == "sw_chunks") w.sw_chunks = value > 0 ? value : 0;
    else if (n == "sw_transpose") w.sw_transpose = value < 0 ? -1 : value > 0 ? 1 : 0;
    else if (n == "sw_no_zero_copy") w.sw_no_zero_copy = value > 0;
    else if (n == "sw_clock") w.sw_clock = value != 0;
    else if (n == "region_sw_all") w.region_sw_all = value < 0 ? -1 : value;
    else if (n == "region_server") w.region_server = value < 0 ? -1 : value > 0 ? 1 : 0;
    else if (n == "server_idle_us") w.server_idle_us = value > 0 ? value : 1;
    else if (n == "server_trace") w.server_trace = value != 0;
    else if (n == "region_flag_wait") w.region_flag_wait = value != 0;
    else if (n == "region_pick_timeout_us") w.region_pick_timeout_us = value > 0 ? value : 1;
    else if (n == "region_debug_pick") w.region_debug_pick = value > 0 ? value : 0;
    else if (n == "mirror_canary") w.mirror_canary = value > 0 ? value : 0;
    else if (n == "route_shared") w.route_shared = value;
    else if (n == "sw_lanes") w.sw_lanes = value == 8 || value == 16 || value == 32 || value == 64 ? value : 0;
    else {
        h->err = "phmm_set_switch: unknown switch";
        return PHMM_ERR_INVALID_ARG;
    }
    h->sw_touched = true;  // (its calls stay on its own resources from now on: route_shared)
    if (h->comb) phmm_host::combiner_set_switches(h->comb, w);
    return PHMM_OK;
}

uint64_t phmm_get_stat(phmm_handle *h, const char *name) {
    if (!h || !name) return 0;
    const std::string n(name);
    uint64_t own = 0;
    if (n == "staged_bytes") own = h->stat_staged_bytes;
    else if (n == "rescue_passes") own = h->stat_rescue_passes;
    else if (n == "sw_kernel_us") return h->swork.last_kernel_us;
    else if (n == "sw_backtrack_bytes") return h->swork.last_backtrack_bytes;
    else if (n == "sw_second_pass") return h->swork.last_second_pass;
    else if (n == "sw_clock_mhz") return h->swork.last_clock_mhz;
    else if (n == "region_sw_all") own = h->swork.region_sw_all_calls;
    else if (n == "region_pick_timeouts") own = h->swork.region_pick_timeouts;
    else if (n.rfind("server_", 0) == 0) return phmm_host::server_stat(h->device, name);  // (the device's server: every handle's calls)
    else return 0;
    return own + (h->comb ? phmm_host::combiner_stat(h->comb, name) : 0);
}

int phmm_plan_describe(unsigned flags, uint32_t concurrent_callers, uint32_t n_regions, const uint32_t *region_read_off,
                       const uint32_t *region_hap_off, const uint32_t *read_off, const uint32_t *hap_off, phmm_plan_info *info) {
    if (!info) return PHMM_ERR_INVALID_ARG;
    try {
        memset(info, 0, sizeof *info);
        std::vector<uint64_t> oo((size_t)n_regions + 1, 0);
        if (region_read_off && region_hap_off)
            for (uint32_t g = 0; g < n_regions; ++g)
                oo[g + 1] = oo[g] + (uint64_t)(region_read_off[g + 1] - region_read_off[g]) * (region_hap_off[g + 1] - region_hap_off[g]);
        phmm_handle h;  // host-only: carries the flags and the planner's switches (this process's PHMM_* environment), never a device
        read_env_switches(h.sw);
        h.flags = flags;
        h.gpu_sharers = concurrent_callers ? concurrent_callers : 1u;
        // (a dry batch owns nothing on a device: plain delete, also when something below throws)
        std::unique_ptr<phmm_batch> owner(batch_create_impl(&h, n_regions, region_read_off, region_hap_off, read_off, hap_off, oo.data(), false, 0, true));
        phmm_batch *b = owner.get();
        if (!b) return h.err_code == PHMM_ERR_NO_MEMORY ? PHMM_ERR_NO_MEMORY : PHMM_ERR_INVALID_ARG;
        info->cells = b->cells;
        info->n_launches = phmm_batch_num_launches(b);
        for (const auto &g : b->chain_groups) {
            info->n_chain_launches += 1;
            info->chain_items += g.items.size();
            uint32_t fewest = 0xffffffffu;
            for (const ChainItem &it : g.items) fewest = std::min(fewest, it.read_end - it.read_begin);
            info->min_reads_per_run = info->min_reads_per_run ? std::min(info->min_reads_per_run, fewest) : fewest;
        }
        for (const auto &c : b->classes)
            if (c.chain) info->chain_cells += c.cells;
        info->swept_cells = b->swept_cells;
        info->pad_column_cells = b->pad_column_cells;
        info->pad_slot_cells = b->pad_slot_cells;
        snprintf(info->dominant_kernel, sizeof info->dominant_kernel, "%s", b->dominant.c_str());
        return PHMM_OK;
    } catch (const std::bad_alloc &) {
        return PHMM_ERR_NO_MEMORY;
    } catch (...) {
        return PHMM_ERR_INTERNAL;
    }
}

uint64_t phmm_batch_cells(const phmm_batch *b) { return b ? b->cells : 0; }
uint64_t phmm_batch_algorithmic_bytes(const phmm_batch *b) { return b ? b->alg_bytes : 0; }
uint32_t phmm_batch_num_launches(const phmm_batch *b) {
    if (!b) return 0;
    uint32_t n = 0;
    for (const auto &g : b->chain_groups) n += g.items.empty() ? 0u : 1u;
    for (const auto &c : b->classes)
        if (!c.chain || c.f32_first) n += 1u;  // per-read classes, and the f64 redo behind an f32 sweep
    return n;
}
const char *phmm_batch_dominant_kernel(const phmm_batch *b) { return b ? b->dominant.c_str() : ""; }

}  // extern "C"
