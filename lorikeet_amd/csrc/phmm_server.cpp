// Host side of the resident region server (phmm_server.hpp, phmm_server_kernels.hip): one server per device, shared by every
// handle and every calling thread of the process.  A region call -- phmm_region_compute on any handle, phmm_region_submit on a
// shared one -- that fits the server's limits
//   1. takes a SLOT (4 MB of pinned host memory with a device arena of the same layout),
//   2. stages its inputs and the job record (the kernels' parameter blocks, made here) into the slot's mirror,
//   3. writes one 64-byte ring entry (and launches the server if none is on the chip),
//   4. polls the finish word the last task stores into the mirror, and copies its results out.
// No launch, no stream synchronisation, no other caller's flush on the way.  The steps are the reference's, in its order
// (haplotype_caller_engine.rs:1311-1357); a call the limits refuse takes the launched pipeline of phmm_region.cpp instead.
// No CPU path: without a device there is no server and no handle.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "phmm_region_internal.hpp"
#include "phmm_server.hpp"
#include "phmm_tables.hpp"

using namespace phmm;
using namespace phmm_host;

namespace {

constexpr size_t kSlotBytes = 4u << 20;   // one slot: staged inputs + device-only pieces + results
constexpr size_t kStageMax = 1u << 20;    // inputs of one job (the stage-in tasks copy them over the link)
constexpr int kMaxSlots = 64;
constexpr int kServerStallMs = 5000;     // calls in flight and none finishing for this long: the server gives up, is not used again by the process,
                                          // and its calls are run again by the launched pipeline
constexpr uint32_t kSwCapacity = 24;      // CIGAR elements per read -> haplotype alignment (a call that needs more takes the launched pipeline)
constexpr uint32_t kMaxHap = 512;          // 16 lanes x 25 columns per pair up to 400 bases, 32 x 16 beyond
const int kSwKs[] = {2, 3, 4, 5, 6, 8};

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

struct Slot {
    char *dev = nullptr, *host = nullptr, *host_dev = nullptr;  // host_dev: the mirror as the device sees it
    char *sync = nullptr;  // device memory that only agent-scope atomics ever touch: [status | group_done x kSyncReads | helper_out]
};
constexpr size_t kSyncReads = 16384;                       // reads of one call at most
constexpr size_t kSyncHelperBytes = 1u << 20;              // ... and reads x haplotypes (rounded up to eight) x 8 bytes of helpers' likelihoods
constexpr size_t kSyncBytes = 256 + 4 * kSyncReads + kSyncHelperBytes;

// The server's own lock: held for a handful of stores (a slot taken or given back, a ring entry written).  A spinning one: a
// std::mutex parked and woke ten callers' threads for critical sections of 100 ns -- staging + publishing took 48 us a call
// instead of 12, handing the results over 20 instead of 1 (tools/server_trace, NOTEBOOK 20.3).
struct SpinLock {
    std::atomic<bool> held{false};
    void lock() {
        for (;;) {
            if (!held.exchange(true, std::memory_order_acquire)) return;
            while (held.load(std::memory_order_relaxed)) __builtin_ia32_pause();
        }
    }
    void unlock() { held.store(false, std::memory_order_release); }
};

struct Server {
    int device = 0;
    SpinLock mu;
    bool ok = false, broken = false;
    std::string why_broken;
    hipStream_t stream = nullptr;
    char *d_block = nullptr;  // [SrvCtl | SrvMail x SRV_MAIL x 2]: zeroed in front of every launch
    size_t block_bytes = 0;
    SrvRegion *d_regions = nullptr;
    SrvEntry *ring = nullptr;
    SrvExit *exit_word = nullptr;   // [0]: what a launch says when it leaves; 128 bytes on: the yield word (host -> dispatcher)
    const SrvEntry *ring_dev = nullptr;
    SrvExit *exit_dev = nullptr;
    uint32_t *d_slab = nullptr;
    size_t slab_stride = 0;  // dwords per worker
    uint32_t n_blocks = 0;
    uint32_t next_seq = 0, consumed = 0, epoch = 0;
    bool running = false;
    Slot slots[kMaxSlots];
    int n_slots = 0;
    std::vector<int> free_slots;
    std::atomic<int> in_flight{0};
    std::atomic<uint32_t> tasks_in_flight{0};  // of the calls in flight: kept below SRV_MAIL_TASKS (phmm_server.hpp)
    std::atomic<uint64_t> n_jobs{0}, n_launches{0};
    std::atomic<uint64_t> ns_stage{0}, ns_wait{0}, ns_out{0};  // host time of the calls so far: staging + publishing, polling, handing the results over
    SrvTrace *d_trace = nullptr;
    uint32_t trace_cap = 0;
};

std::mutex g_mu;
Server *g_servers[kMaxDevices] = {};

bool hip_ok(hipError_t e) {
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

// (g_mu held) everything a device's server owns but its slots; nullptr when the device refuses
Server *server_create(int device) {
    DevGuard dg(device);
    Server *S = new Server();
    S->device = device;
    const int per_cu = server_blocks_per_cu();
    int cus = 0;
    if (per_cu <= 0 || !hip_ok(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) || cus <= 0) return S;
    S->n_blocks = (uint32_t)per_cu * (uint32_t)cus;
    // (the aligner's flags: one slab per worker wave, for the longest sweep the limits admit)
    S->slab_stride = (size_t)(std::max<uint32_t>(kMaxHap, SRV_MAX_ROWS) + 64) * (size_t)sw_flag_words(8) * 64;
    S->block_bytes = sizeof(SrvCtl) + 2 * sizeof(SrvMail) * SRV_MAIL;
    void *ring_dev = nullptr, *exit_dev = nullptr;
    const bool good = hip_ok(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking)) && hip_ok(hipMalloc((void **)&S->d_block, S->block_bytes)) &&
                      hip_ok(hipMalloc((void **)&S->d_regions, sizeof(SrvRegion) * SRV_RING)) &&
                      hip_ok(hipMalloc((void **)&S->d_slab, S->slab_stride * 4 * S->n_blocks)) &&
                      hip_ok(hipHostMalloc((void **)&S->ring, sizeof(SrvEntry) * SRV_RING, hipHostMallocDefault)) &&
                      hip_ok(hipHostMalloc((void **)&S->exit_word, 256, hipHostMallocDefault)) &&
                      hip_ok(hipHostGetDevicePointer(&ring_dev, S->ring, 0)) && hip_ok(hipHostGetDevicePointer(&exit_dev, S->exit_word, 0));
    if (!good) return S;
    memset(S->ring, 0, sizeof(SrvEntry) * SRV_RING);
    memset(S->exit_word, 0, 256);
    S->ring_dev = (const SrvEntry *)ring_dev;
    S->exit_dev = (SrvExit *)exit_dev;
    S->ok = true;
    return S;
}

Server *server_of(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    Server *&S = g_servers[device % kMaxDevices];
    if (!S) S = server_create(device);
    return S;
}

// (S.mu held) has the launch the host believes to be running said good-bye?
void observe_exit(Server &S) {
    if (!S.running) return;
    if (__atomic_load_n(&S.exit_word->epoch, __ATOMIC_ACQUIRE) != S.epoch) return;
    S.running = false;
    S.consumed = S.exit_word->consumed;
    if (S.exit_word->fault && !S.broken) {
        S.broken = true;
        S.why_broken = "the region server made no progress within its time limit";
    }
}

// (S.mu held) a launch that starts at ring entry S.consumed.  Not while region calls of the launched kind are in flight: the
// server's waves fill every SIMD, and kernels launched beside them would wait for it to leave -- the calls that are waiting
// for the server look again (server_region_wait) and launch it when those are through.
bool launch_locked(Server &S, const Switches &sw) {
    if (region_calls_in_flight(S.device) > 0) return true;
    DevGuard dg(S.device);
    if (sw.server_trace && !S.d_trace) {
        S.trace_cap = 1u << 20;
        if (!hip_ok(hipMalloc((void **)&S.d_trace, sizeof(SrvTrace) * S.trace_cap))) S.d_trace = nullptr;
    }
    SrvParams P{};
    P.ctl = (SrvCtl *)S.d_block;
    P.mail = (SrvMail *)(S.d_block + sizeof(SrvCtl));
    P.regions = S.d_regions;
    P.ring = S.ring_dev;
    P.exit_word = S.exit_dev;
    P.yield_word = (const uint32_t *)((const char *)S.exit_dev + 128);
    P.start_seq = S.consumed;
    P.epoch = S.epoch + 1;
    P.idle_ticks = 100u * (uint32_t)std::max(1, sw.server_idle_us);
    P.stall_ticks = 100u * 1000u * (uint32_t)kServerStallMs;
    P.trace = sw.server_trace ? S.d_trace : nullptr;
    P.trace_cap = S.trace_cap;
    if (!hip_ok(hipMemsetAsync(S.d_block, 0, S.block_bytes, S.stream)) || !hip_ok(launch_server(P, S.n_blocks, S.stream))) {
        S.broken = true;
        S.why_broken = "the region server could not be launched";
        return false;
    }
    S.epoch += 1;
    S.running = true;
    S.n_launches.fetch_add(1, std::memory_order_relaxed);
    return true;
}

// a slot of the server for one call (-1: every slot taken, or no memory for another)
int take_slot(Server &S) {
    std::lock_guard<SpinLock> lk(S.mu);
    if (!S.free_slots.empty()) {
        const int slot = S.free_slots.back();
        S.free_slots.pop_back();
        return slot;
    }
    if (S.n_slots >= kMaxSlots) return -1;
    DevGuard dg(S.device);
    Slot &T = S.slots[S.n_slots];
    void *dp = nullptr;
    if (hip_ok(hipMalloc((void **)&T.dev, kSlotBytes)) && hip_ok(hipMalloc((void **)&T.sync, kSyncBytes)) &&
        hip_ok(hipHostMalloc((void **)&T.host, kSlotBytes, hipHostMallocDefault)) && hip_ok(hipHostGetDevicePointer(&dp, T.host, 0)) && dp) {
        T.host_dev = (char *)dp;
        return S.n_slots++;
    }
    if (T.dev) (void)hipFree(T.dev);
    if (T.sync) (void)hipFree(T.sync);
    if (T.host) (void)hipHostFree(T.host);
    T = Slot();
    return -1;
}

int fail(std::string *err, const std::string &msg, int code) {
    if (err) *err = msg;
    return code;
}

}  // namespace

namespace phmm_host {

struct ServerPending {
    Server *S = nullptr;
    int slot = -1;
    uint32_t seq = 0;
    RegionArgs a;
    Layout L;
    uint64_t n_out = 0;
    uint32_t n_tasks = 0;
    std::chrono::steady_clock::time_point t0;
};

void server_pending_free(ServerPending *p) { delete p; }

}  // namespace phmm_host

namespace {
// The staged call `p` (its slot, its task counts) goes into the ring; the server is launched if none is on the chip.  false: the
// server is unusable -- slot and counts are given back, the caller takes the launched pipeline.
bool publish(Server &S, const Switches &sw, phmm_host::ServerPending *p, const uint32_t *n_tasks, uint32_t n16, uint32_t job_off) {
    const Slot &T = S.slots[p->slot];
    S.in_flight.fetch_add(1, std::memory_order_relaxed);
    std::lock_guard<SpinLock> lk(S.mu);
    observe_exit(S);
    auto give_back = [&] {
        S.free_slots.push_back(p->slot);
        S.in_flight.fetch_sub(1, std::memory_order_relaxed);
        S.tasks_in_flight.fetch_sub(p->n_tasks, std::memory_order_relaxed);
        return false;
    };
    if (S.broken) return give_back();
    const uint32_t seq = S.next_seq++;
    p->seq = seq;
    SrvEntry &e = S.ring[seq & (SRV_RING - 1)];
    e.slot = (uint32_t)p->slot;
    for (uint32_t k = 0; k < SRV_KINDS; ++k) e.n[k] = n_tasks[k];
    e.flags = sw.server_trace ? 1u : 0u;
    e.stage_n16 = n16;
    e.job_off = job_off;
    e.stage_src = (uint64_t)(uintptr_t)T.host_dev;
    e.stage_dst = (uint64_t)(uintptr_t)T.dev;
    __atomic_store_n(&e.valid, seq + 1u, __ATOMIC_RELEASE);
    if (!S.running && !launch_locked(S, sw)) {
        // (nothing on the chip will ever look at the entry: take it back)
        __atomic_store_n(&e.valid, 0u, __ATOMIC_RELEASE);
        S.next_seq = seq;
        return give_back();
    }
    return true;
}
}  // namespace

namespace phmm_host {

// Stage one call in a slot and hand it to the device's server.  kServerNotTaken: the call is outside the server's limits (or
// the server is not to be used): nothing was done, the caller takes the launched pipeline.  `a` is validated.
constexpr int kServerFromHandles = 5;

int server_region_submit(phmm_handle *h, const RegionArgs &a, ServerPending **out, bool via_submit) {
    *out = nullptr;
    const auto t_enter = std::chrono::steady_clock::now();
    // Which calls: with no switch set, the one-shot region calls of PRIVATE handles once more than kServerFromHandles of the
    // caller's handles are alive on the device -- a handle per worker thread at Lorikeet's --threads 10.  The launched pipelines of
    // many handles share the command processor's pipes and every caller sits in its own chain of launches: 22 k regions/s
    // (128 x 8) from four callers on, whatever their number; a call through the server is ~260 us with few callers (its waiters
    // spin while cores are free), so N callers get N / 260 us: the server overtakes at six callers (23.3 against 22.1 k; 30 x 3
    // regions at five, the ragged mix at six to seven -- profiles/r06_server_threshold.txt), and its results are the region's own
    // bits whatever the load.  Up to five handles keep their queues (faster there), a shared handle's phmm_region_submit keeps its
    // combiner (faster, and it says that it combines).
    if (h->sw.region_server == 0) return kServerNotTaken;
    if (h->sw.region_server < 0 && (via_submit || h->sw_touched || h->internal || h->comb || h->sw.route_shared > 0 || user_handles_on(h->device) <= kServerFromHandles))
        return kServerNotTaken;  // (route_shared > 0: the caller asked for the combiner instead)
    const uint32_t ng = a.n_regions, nr = a.region_read_off[ng], nh = a.region_hap_off[ng];
    if (!ng || !nr || !nh) return kServerNotTaken;
    if (ng >= 8 && (size_t)a.read_off[nr] > one_shot_bytes()) return kServerNotTaken;  // (large batches: the chunk pipeline)
    // ---- limits ------------------------------------------------------------------------------------------------------------
    uint32_t max_r = 0, max_h = 0, max_nh = 0, max_hap_cigar = 0;
    for (uint32_t g = 0; g < ng; ++g) {
        const uint32_t nrg = a.region_read_off[g + 1] - a.region_read_off[g], nhg = a.region_hap_off[g + 1] - a.region_hap_off[g];
        if (!nrg || !nhg) return kServerNotTaken;
        max_nh = std::max(max_nh, nhg);
        if (a.out_off[g + 1] - a.out_off[g] != (uint64_t)nrg * nhg) return kServerNotTaken;  // (results lie back to back)
    }
    if (a.out_off[0] != 0) return kServerNotTaken;
    for (uint32_t r = 0; r < nr; ++r) {
        const uint32_t len = a.read_off[r + 1] - a.read_off[r];
        if (!len) return kServerNotTaken;
        max_r = std::max(max_r, len);
    }
    for (uint32_t x = 0; x < nh; ++x) {
        const uint32_t len = a.hap_off[x + 1] - a.hap_off[x];
        if (!len) return kServerNotTaken;
        max_h = std::max(max_h, len);
        max_hap_cigar = std::max(max_hap_cigar, a.hap_cigar_off[x + 1] - a.hap_cigar_off[x]);
    }
    if (max_r > SRV_MAX_ROWS || max_h > kMaxHap) return kServerNotTaken;
    // (a result below -600 needs the exact pass, which only the launched pipeline carries: phmm_region.cpp, "can_underflow")
    if (a.cfg.constant_gcp == 0 || 53.0 + (double)max_r * a.cfg.constant_gcp / 10.0 >= 590.0 || h->sw.no_rescue) return kServerNotTaken;
    {   // the aligner's scaled instances only (phmm_sw.cpp, sw_plan)
        const phmm_sw_parameters &w = a.rcfg.sw_parameters;
        const int64_t big = std::max(std::max(std::llabs((long long)w.match_value), std::llabs((long long)w.mismatch_penalty)),
                                     std::max(std::llabs((long long)w.gap_open_penalty), std::llabs((long long)w.gap_extend_penalty)));
        if (big * ((int64_t)max_h + max_r + 2) >= 100000000) return kServerNotTaken;
    }
    // (the sweep's lane geometry is a function of the call's own longest haplotype: the same bits under any load)
    const uint32_t fwd_l = max_h <= 16u * SRV_MAX_K ? 16u : 32u, group_haps = 64 / fwd_l;
    const uint32_t fwd_k = std::max<uint32_t>(fwd_l == 16 ? 2 : 13, (max_h + fwd_l - 1) / fwd_l);
    uint32_t sw_k = 0;
    for (int k : kSwKs)
        if (!sw_k && (uint32_t)k * 64 >= max_h) sw_k = (uint32_t)k;
    const size_t lds_ref = (max_h + 15) / 16 * 16, lds_alt = (max_r + 15) / 16 * 16;
    const size_t lds_group = (lds_ref + lds_alt + 4ull * (max_r + 1) + 15) / 16 * 16;
    const uint32_t pj_capacity = 4 * (kSwCapacity + max_hap_cigar + 2) + 8;
    const uint32_t prep_rows = (max_r + 1 + 7) / 8 * 8;
    if (!sw_k || lds_group > SRV_LDS_BYTES || 16ull * pj_capacity > SRV_LDS_BYTES || (size_t)prep_rows * 17 > SRV_LDS_BYTES) return kServerNotTaken;
    if (!h->server) h->server = server_of(h->device);  // (the device's one server: looked up once per handle)
    Server *S = (Server *)h->server;
    if (!S || !S->ok || S->broken) return kServerNotTaken;
    if ((size_t)(std::max(max_h, max_r) + 64) * (size_t)sw_flag_words((int)sw_k) * 64 > S->slab_stride) return kServerNotTaken;
    // ---- the job's layout in a slot: [offset arrays | job record | inputs ... status] staged, then device-only, then results ------
    size_t used = 0;
    auto take = [&](size_t bytes) {
        const size_t off = up256(used);
        used = off + bytes;
        return off;
    };
    const size_t o_read_region = take(4ull * nr), o_rro = take(4ull * (ng + 1)), o_rho = take(4ull * (ng + 1)), o_ro = take(4ull * (nr + 1)),
                 o_ho = take(4ull * (nh + 1)), o_oo = take(8ull * (ng + 1)), o_job = take(sizeof(SrvJob));
    const Layout L(up256(used), a, kSwCapacity, 0);
    // (what waves of different XCDs hand each other lies in a buffer of the slot's own that only agent-scope atomics ever touch)
    const size_t helper_stride = (max_nh + 7) / 8 * 8;
    if (L.in_end > kStageMax || L.end > kSlotBytes || nr > kSyncReads || 8ull * nr * helper_stride > kSyncHelperBytes) return kServerNotTaken;
    // ---- the tasks ---------------------------------------------------------------------------------------------------------------------
    const uint32_t n16 = (uint32_t)((L.in_end + 15) / 16), groups = (max_nh + group_haps - 1) / group_haps;
    uint32_t n_tasks[SRV_KINDS];
    n_tasks[SRV_STAGE] = (n16 + SRV_STAGE_UNITS - 1) / SRV_STAGE_UNITS;
    n_tasks[SRV_CHAIN] = nr * groups;  // (a wave per read and group of four haplotypes; the group-0 wave goes on to the read's alignment)
    uint32_t total_tasks = 0;
    for (uint32_t k = 0; k < SRV_KINDS; ++k) total_tasks += n_tasks[k];
    // (every task posted waits in a mailbox of its own until a worker takes it: the calls in flight must not have more tasks than
    // there are mailboxes to go round)
    if (S->tasks_in_flight.fetch_add(total_tasks, std::memory_order_relaxed) + total_tasks > SRV_MAIL_TASKS) {
        S->tasks_in_flight.fetch_sub(total_tasks, std::memory_order_relaxed);
        return kServerNotTaken;
    }
    const int slot = take_slot(*S);
    if (slot < 0) {  // (every slot taken, or no memory for another: the launched pipeline)
        S->tasks_in_flight.fetch_sub(total_tasks, std::memory_order_relaxed);
        return kServerNotTaken;
    }
    const Slot &T = S->slots[slot];
    char *const hs = T.host, *const dev = T.dev, *const mirror = T.host_dev;
    // ---- inputs into the mirror ------------------------------------------------------------------------------------------------------
    auto put = [&](size_t at, const void *src, size_t bytes) {
        if (src && bytes) memcpy(hs + at, src, bytes);
    };
    {
        uint32_t *rr = (uint32_t *)(hs + o_read_region);
        for (uint32_t g = 0; g < ng; ++g)
            for (uint32_t r = a.region_read_off[g]; r < a.region_read_off[g + 1]; ++r) rr[r] = g;
    }
    put(o_rro, a.region_read_off, 4ull * (ng + 1));
    put(o_rho, a.region_hap_off, 4ull * (ng + 1));
    put(o_ro, a.read_off, 4ull * (nr + 1));
    put(o_ho, a.hap_off, 4ull * (nh + 1));
    put(o_oo, a.out_off, 8ull * (ng + 1));
    const size_t rb = a.read_off[nr], hb = a.hap_off[nh];
    put(L.bases, a.read_bases, rb);
    put(L.q0, a.base_q, rb);
    put(L.i0, a.ins_q, rb);
    put(L.d0, a.del_q, rb);
    put(L.mapq, a.mapq, nr);
    put(L.haps, a.hap_bases, hb);
    put(L.refhap, a.region_ref_hap, 4ull * ng);
    put(L.pri, a.hap_priority, 4ull * nh);
    put(L.rstart, a.region_reference_start, 8ull * ng);
    put(L.hco, a.hap_cigar_off, 4ull * (nh + 1));
    put(L.hc, a.hap_cigar, 4ull * a.hap_cigar_off[nh]);
    put(L.hs, a.hap_start_wrt_ref, 4ull * nh);
    put(L.oco, a.orig_cigar_off, 4ull * (nr + 1));
    put(L.oc, a.orig_cigar, 4ull * a.orig_cigar_off[nr]);
    put(L.outco, a.out_cigar_off, 8ull * (nr + 1));
    put(L.clip, a.read_soft_clip, 8ull * nr);
    memset(hs + L.status_in, 0, 256);
    memset(hs + L.res, 0, 256);
    __atomic_fetch_add(&h->stat_staged_bytes, (uint64_t)((2 + (a.ins_q ? 1 : 0) + (a.del_q ? 1 : 0)) * rb + hb), __ATOMIC_RELAXED);  // (a shared handle: many threads)
    // ---- the job record: what region_enqueue (phmm_region.cpp) hands its launches, made once for the tasks ------------------------------
    const uint32_t *d_read_region = (const uint32_t *)(dev + o_read_region), *d_rro = (const uint32_t *)(dev + o_rro), *d_rho = (const uint32_t *)(dev + o_rho),
                   *d_ro = (const uint32_t *)(dev + o_ro), *d_ho = (const uint32_t *)(dev + o_ho);
    const uint64_t *d_oo = (const uint64_t *)(dev + o_oo);
    SrvJob *job = new (hs + o_job) SrvJob();
    {
        PrepParams &pp = job->prep;
        pp.n_reads = nr;
        pp.read_off = d_ro;
        pp.read_bases = (const uint8_t *)(dev + L.bases);
        pp.base_q = (const uint8_t *)(dev + L.q0);
        pp.ins_q = a.ins_q ? (const uint8_t *)(dev + L.i0) : nullptr;
        pp.del_q = a.del_q ? (const uint8_t *)(dev + L.d0) : nullptr;
        pp.mapq = (const uint8_t *)(dev + L.mapq);
        pp.pcr_cache = a.cfg.pcr_error_model ? h->d_pcr_cache + 128 * a.cfg.pcr_error_model : nullptr;
        pp.out_q = (uint8_t *)(dev + L.q);
        pp.out_ins = (uint8_t *)(dev + L.i);
        pp.out_del = (uint8_t *)(dev + L.d);
        pp.out_gcp = (uint8_t *)(dev + L.g);
        pp.threshold = (double *)(dev + L.thr);
        pp.lds_rows = prep_rows;
        pp.waves_per_read = 1;  // (the chain's wave takes the whole read)
        pp.default_indel_qual = 45;  // ReadUtils::DEFAULT_INSERTION_DELETION_QUAL (read_utils.rs:23)
        pp.constant_gcp = a.cfg.constant_gcp;
        pp.base_quality_score_threshold = a.cfg.base_quality_score_threshold;
        pp.disable_cap_to_mapq = a.cfg.disable_cap_read_qualities_to_mapq;
        pp.dynamic_disqualification = a.cfg.dynamic_read_disqualification;
        pp.read_disqualification_scale = a.cfg.read_disqualification_scale;
        pp.expected_error_rate_per_base = a.cfg.expected_error_rate_per_base;
    }
    {
        ForwardParams &f = job->fwd;
        f.n_items = nr;
        f.read_region = d_read_region;
        f.region_read_off = d_rro;
        f.region_hap_off = d_rho;
        f.read_off = d_ro;
        f.hap_off = d_ho;
        f.out_off = d_oo;
        f.read_bases = (const uint8_t *)(dev + L.bases);
        f.base_q = (const uint8_t *)(dev + L.q);
        f.ins_q = (const uint8_t *)(dev + L.i);
        f.del_q = (const uint8_t *)(dev + L.d);
        f.gcp = (const uint8_t *)(dev + L.g);
        f.hap_bases = (const uint8_t *)(dev + L.haps);
        f.out = (double *)(dev + L.out);
        f.eps = h->d_eps;
        f.eps_mis = h->d_eps_mis;
        f.mm = h->d_mm;
        f.ratio_mis = h->d_ratio_mis;
        f.inv_om = h->d_inv_om;
        f.initial_condition = initial_condition();
        f.initial_condition_log10 = initial_condition_log10();
        f.status = (uint32_t *)T.sync;
    }
    {
        PostParams &po = job->pb.post;
        po.n_reads = nr;
        po.read_region = d_read_region;
        po.region_read_off = d_rro;
        po.region_hap_off = d_rho;
        po.out_off = d_oo;
        po.region_ref_hap = (const int32_t *)(dev + L.refhap);
        po.out = (double *)(dev + L.out);
        po.out_final = (double *)(mirror + L.out);
        po.threshold = (const double *)(dev + L.thr);
        po.keep = (uint8_t *)(dev + L.keep);
        po.status_in = nullptr;
        po.status_out = nullptr;  // (the call's last wave hands the status word on: SrvJob::status_out)
        po.max_likelihood_difference_cap = a.cfg.log10_global_read_mismapping_rate;
        po.symmetric = a.cfg.symmetrically_normalize_alleles_to_reference;
        BestParams &bp = job->pb.best;
        bp.r_begin = 0;
        bp.n_reads = nr;
        bp.n_regions = ng;
        bp.region_read_off = d_rro;
        bp.region_hap_off = d_rho;
        bp.out_off = d_oo;
        bp.likelihoods = (const double *)(dev + L.out);
        bp.keep = (const uint8_t *)(dev + L.keep);
        bp.priority = a.hap_priority ? (const int32_t *)(dev + L.pri) : nullptr;
        bp.threshold = a.rcfg.informative_threshold;
        bp.best_allele = (int32_t *)(mirror + L.best);
        bp.likelihood = (double *)(mirror + L.lk);
        bp.confidence = (double *)(mirror + L.conf);
        bp.ref_index = (uint32_t *)(dev + L.refidx);
        job->pb.skip_single_allele = (a.rcfg.flags & PHMM_REGION_SKIP_SINGLE_ALLELE) ? 1u : 0u;
        job->pb.keep_final = (uint8_t *)(mirror + L.keep);
    }
    {
        SwParams &sp = job->sw;
        sp.a_begin = 0;
        sp.n_alignments = nr;
        sp.ref_off = d_ho;
        sp.alt_off = d_ro;
        sp.ref_index = (const uint32_t *)(dev + L.refidx);
        sp.ref_bases = (const uint8_t *)(dev + L.haps);
        sp.alt_bases = (const uint8_t *)(dev + L.bases);
        sp.w_match = a.rcfg.sw_parameters.match_value;
        sp.w_mismatch = a.rcfg.sw_parameters.mismatch_penalty;
        sp.w_open = a.rcfg.sw_parameters.gap_open_penalty;
        sp.w_extend = a.rcfg.sw_parameters.gap_extend_penalty;
        sp.strategy = a.rcfg.overhang_strategy;
        sp.cigar_off = nullptr;
        sp.cigar_slot = kSwCapacity;
        sp.alt_clip = a.read_soft_clip ? (const uint32_t *)(dev + L.clip) : nullptr;
        sp.cigar = (uint32_t *)(dev + L.swc);
        sp.n_cigar = (uint32_t *)(dev + L.nsw);
        sp.alignment_offset = (int32_t *)(dev + L.swo);
        sp.slab = S->d_slab;
        sp.slab_stride = S->slab_stride;
        sp.status = (uint32_t *)(mirror + L.res + 64);
        sp.max_ref = max_h;
        sp.max_alt = max_r;
        sp.lds_ref_bytes = (uint32_t)lds_ref;
        sp.lds_alt_bytes = (uint32_t)lds_alt;
        sp.lds_group_bytes = (uint32_t)lds_group;
        sp.groups_per_block = 1;
        sp.read_region = d_read_region;
        sp.region_hap_off = d_rho;
    }
    {
        ProjectParams &pj = job->pj;
        pj.r_begin = 0;
        pj.n_reads = nr;
        pj.n_regions = ng;
        pj.region_read_off = d_rro;
        pj.region_hap_off = d_rho;
        pj.read_off = d_ro;
        pj.read_bases = (const uint8_t *)(dev + L.bases);
        pj.hap_off = d_ho;
        pj.hap_bases = (const uint8_t *)(dev + L.haps);
        pj.region_ref_hap = (const int32_t *)(dev + L.refhap);
        pj.region_reference_start = (const uint64_t *)(dev + L.rstart);
        pj.hap_cigar_off = (const uint32_t *)(dev + L.hco);
        pj.hap_cigar = (const uint32_t *)(dev + L.hc);
        pj.hap_start_wrt_ref = (const uint32_t *)(dev + L.hs);
        pj.best_allele = nullptr;  // (derived from ref_index)
        pj.ref_index = (const uint32_t *)(dev + L.refidx);
        pj.sw_cigar_off = nullptr;
        pj.sw_cigar_slot = kSwCapacity;
        pj.sw_pair_stride = 0;
        pj.sw_cigar = (const uint32_t *)(dev + L.swc);
        pj.n_sw_cigar = (const uint32_t *)(dev + L.nsw);
        pj.sw_offset = (const int32_t *)(dev + L.swo);
        pj.read_clip = a.read_soft_clip ? (const uint32_t *)(dev + L.clip) : nullptr;
        pj.orig_cigar_off = (const uint32_t *)(dev + L.oco);
        pj.orig_cigar = (const uint32_t *)(dev + L.oc);
        pj.out_cigar_off = (const uint64_t *)(dev + L.outco);
        pj.out_cigar = (uint32_t *)(mirror + L.pout);
        pj.n_out_cigar = (uint32_t *)(mirror + L.pno);
        pj.new_pos = (int64_t *)(mirror + L.pos);
        pj.status = (int32_t *)(mirror + L.pst);
        pj.flags = (uint32_t *)(mirror + L.res + 128);
        pj.workspace = nullptr;  // (the lanes' builders live in the worker wave's LDS)
        pj.capacity = pj_capacity;
    }
    job->fwd_k = fwd_k;
    job->group_haps = group_haps;
    job->groups = groups;
    job->sw_k = sw_k;
    job->n_reads = nr;
    job->group_done = (uint32_t *)(T.sync + 256);
    job->helper_out = (double *)(T.sync + 256 + 4 * kSyncReads);
    job->helper_stride = (uint32_t)helper_stride;
    job->status_in = (uint32_t *)T.sync;
    job->status_out = (uint32_t *)(mirror + L.res);
    job->wait_ticks = 100u * 1000u * (uint32_t)kServerStallMs;
    job->finish_flag = (uint32_t *)(mirror + L.res + 224);
    // ---- the ring entry ----------------------------------------------------------------------------------------------------------------
    ServerPending *p = new ServerPending();
    p->S = S;
    p->slot = slot;
    p->a = a;
    p->L = L;
    p->n_out = a.out_off[ng];
    p->n_tasks = total_tasks;
    p->t0 = std::chrono::steady_clock::now();
    if (!publish(*S, h->sw, p, n_tasks, n16, (uint32_t)o_job)) {
        delete p;
        return kServerNotTaken;
    }
    S->n_jobs.fetch_add(1, std::memory_order_relaxed);
    S->ns_stage.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_enter).count(), std::memory_order_relaxed);
    *out = p;
    return PHMM_OK;
}

// (the calling thread) wait for the finish word of a submitted call; false: the server broke down meanwhile
bool poll_finish(phmm_handle *h, ServerPending *p, const uint32_t *flag) {
    Server &S = *p->S;
    bool done = false;
    // How to wait (tools/ab/waiters.sh, one box): while the calls in flight leave two of the process' cores free the waiter SPINS --
    // a nap of 20 us is 70 with the timer's slack, a tenth of a call: 8 callers 26.9 -> 30.0 k regions/s, 10 callers 32.8 -> 33.4 k,
    // 30 x 3 regions from 10 callers 64 -> 78 k, the ragged mix 26.3 -> 28.5 k; with more calls in flight than that a short spin
    // and then naps between looks (spinning waiters beyond the cores are throttled together with the callers that stage: 32
    // callers 22.9 k spinning, 45.6 k napping; 16: the same either way).
    constexpr uint32_t wait_spins = 64;
    const bool naps = S.in_flight.load(std::memory_order_relaxed) > process_cores() - 2;
    const auto give_up = p->t0 + std::chrono::milliseconds(kServerStallMs * 4);
    if (h->sw.region_debug_pick & 4) {  // (tests: this call's answer counts as lost -- what a stalled server looks like from here)
        std::lock_guard<SpinLock> lk(S.mu);
        S.broken = true;
        S.why_broken = "tests: region_debug_pick & 4";
        return false;
    }
    for (uint32_t spins = 0; !done; ++spins) {
        done = __atomic_load_n(flag, __ATOMIC_ACQUIRE) != 0;
        if (done) break;
        if ((spins & 63u) == 63u) {
            // (is no server on the chip -- it left while this call was on its way, or its launch was put off?  then whoever notices
            // first starts one)
            if (!__atomic_load_n(&S.running, __ATOMIC_RELAXED) || __atomic_load_n(&S.exit_word->epoch, __ATOMIC_ACQUIRE) == __atomic_load_n(&S.epoch, __ATOMIC_RELAXED)) {
                std::lock_guard<SpinLock> lk(S.mu);
                observe_exit(S);
                if (S.broken) break;
                if (!S.running && (int32_t)(p->seq - S.consumed) >= 0 && !launch_locked(S, h->sw)) break;
            }
            if (spins > wait_spins) {
                if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() >= give_up) {
                    std::lock_guard<SpinLock> lk(S.mu);
                    S.broken = true;
                    S.why_broken = "a call did not come back from the region server";
                    break;
                }
                if (naps) std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
        }
        __builtin_ia32_pause();
    }
    return done;
}

// Wait for a submitted call, hand its results to the caller, give the slot back.  kServerRedo: the call has to be run again
// by the launched pipeline (a CIGAR outgrew its slot; *redo holds its arguments); everything else is the call's own status,
// with its message in *err (the caller files it: a private handle's err, a shared handle's per-thread message).
int server_region_wait(phmm_handle *h, ServerPending *p, std::string *err, RegionArgs *redo_args) {
    Server &S = *p->S;
    const Slot &T = S.slots[p->slot];
    const Layout &L = p->L;
    const RegionArgs &a = p->a;
    const uint32_t ng = a.n_regions, nr = a.region_read_off[ng];
    const char *hs = T.host;
    const uint32_t *flag = (const uint32_t *)(hs + L.res + 224);
    int st = PHMM_OK;
    const auto t_enter = std::chrono::steady_clock::now();
    const bool done = poll_finish(h, p, flag);
    if (!done) {
        // The server gave up (nothing finished within its time limit: its waves starved behind long launched kernels, or worse) or
        // never answered.  It is not used again by this process (phmm_get_stat "server_broken"); the slot is not given back -- a
        // kernel may still be writing into it -- and the call is run again by the launched pipeline.
        S.in_flight.fetch_sub(1, std::memory_order_relaxed);
        S.tasks_in_flight.fetch_sub(p->n_tasks, std::memory_order_relaxed);
        if (redo_args) *redo_args = a;
        delete p;
        return kServerRedo;
    }
    const auto t_done = std::chrono::steady_clock::now();
    S.ns_wait.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_done - t_enter).count(), std::memory_order_relaxed);
    const uint32_t *sw_st = (const uint32_t *)(hs + L.res + 64);
    bool redo = false;
    if (sw_st[SW_STATUS_CAPACITY]) {
        redo = true;  // (the launched pipeline grows the alignments' slots and runs the call again)
        if (redo_args) *redo_args = a;
    } else {
        memcpy(a.keep, hs + L.keep, nr);
        if (p->n_out) memcpy(a.out, hs + L.out, 8ull * p->n_out);
        memcpy(a.best_allele, hs + L.best, 4ull * nr);
        memcpy(a.likelihood, hs + L.lk, 8ull * nr);
        memcpy(a.confidence, hs + L.conf, 8ull * nr);
        memcpy(a.status, hs + L.pst, 4ull * nr);
        memcpy(a.n_out_cigar, hs + L.pno, 4ull * nr);
        memcpy(a.new_pos, hs + L.pos, 8ull * nr);
        if (a.out_cigar_off[nr]) memcpy(a.out_cigar, hs + L.pout, 4ull * a.out_cigar_off[nr]);
        if (status_positive(*(const uint32_t *)(hs + L.res), false)) {
            st = fail(err, "PairHmm Log Probability cannot be greater than 0.0", PHMM_ERR_POSITIVE_RESULT);  // pair_hmm.rs:478-481
        } else if (sw_st[SW_STATUS_EMPTY]) {  // the reference asserts (smith_waterman_aligner.rs:65-68, :132-134)
            st = fail(err, "phmm_region_compute: non-empty sequences are required for the Smith-Waterman calculation", PHMM_ERR_INVALID_ARG);
        } else if (((const uint32_t *)(hs + L.res + 128))[1]) {  // (a main wave gave up waiting for a helper)
            st = fail(err, "phmm_region_compute: internal error, a wait inside the region server ran out of time", PHMM_ERR_INTERNAL);
        } else if (*(const uint32_t *)(hs + L.res + 128) & 1u) {
            st = fail(err, "phmm_region_compute: a CIGAR needs more elements than its slot holds (n_out_cigar has the sizes)", PHMM_ERR_CIGAR_CAPACITY);
        }
    }
    {
        std::lock_guard<SpinLock> lk(S.mu);
        S.free_slots.push_back(p->slot);
    }
    S.ns_out.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_done).count(), std::memory_order_relaxed);
    S.in_flight.fetch_sub(1, std::memory_order_relaxed);
    S.tasks_in_flight.fetch_sub(p->n_tasks, std::memory_order_relaxed);
    delete p;
    return redo ? kServerRedo : st;
}

// A region call of the launched kind is about to enqueue its kernels: if the server is on the chip it is asked to leave (it
// finishes what it has and goes; its next launch waits until no such call is in flight, launch_locked).
void server_yield(int device) {
    Server *S;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        S = g_servers[device % kMaxDevices];
    }
    if (!S || !S->ok || !__atomic_load_n(&S->running, __ATOMIC_RELAXED)) return;
    __atomic_fetch_add((uint32_t *)((char *)S->exit_word + 128), 1u, __ATOMIC_RELEASE);
}

uint64_t server_stat(int device, const char *name) {
    std::lock_guard<std::mutex> lk(g_mu);
    Server *S = g_servers[device % kMaxDevices];
    if (!S) return 0;
    const std::string n(name);
    if (n == "server_jobs") return S->n_jobs.load();
    if (n == "server_launches") return S->n_launches.load();
    if (n == "server_broken") return S->broken ? 1 : 0;
    if (n == "server_stage_ns") return S->ns_stage.load();
    if (n == "server_wait_ns") return S->ns_wait.load();
    if (n == "server_out_ns") return S->ns_out.load();
    return 0;
}

// The last of the caller's handles on the device is going: nothing of the server may be on the chip afterwards (it leaves by
// itself once idle; this waits for that).  Its memory stays with the process.
void server_quiesce(int device) {
    Server *S;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        S = g_servers[device % kMaxDevices];
    }
    if (!S || !S->ok || S->broken) return;
    DevGuard dg(device);
    (void)hipStreamSynchronize(S->stream);
    std::lock_guard<SpinLock> lk(S->mu);
    observe_exit(*S);
}

// The trace of the tasks run so far (developer runs, switch server_trace): up to `cap` records into `out`; returns how many exist.
uint32_t server_trace_read(int device, phmm::SrvTrace *out, uint32_t cap) {
    Server *S;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        S = g_servers[device % kMaxDevices];
    }
    if (!S || !S->ok || !S->d_trace) return 0;
    DevGuard dg(device);
    (void)hipStreamSynchronize(S->stream);
    uint32_t n = 0;
    const SrvCtl *ctl = (const SrvCtl *)S->d_block;
    if (!hip_ok(hipMemcpy(&n, &ctl->trace_count, 4, hipMemcpyDeviceToHost))) return 0;
    const uint32_t have = std::min(n, S->trace_cap);
    if (out && cap && have && !hip_ok(hipMemcpy(out, S->d_trace, sizeof(SrvTrace) * std::min(have, cap), hipMemcpyDeviceToHost))) return 0;
    return have;
}

}  // namespace phmm_host

static_assert(sizeof(SrvTrace) == 72, "trace record as include/phmm.h describes it");
extern "C" uint32_t phmm_server_trace(int device_id, void *out, uint32_t cap) {
    if (device_id < 0 || device_id >= kMaxDevices) return 0;
    return phmm_host::server_trace_read(device_id, (SrvTrace *)out, cap);
}
