// gfx950: the resident region server (phmm_server.hpp) -- one kernel of one-wave blocks that stays on the chip while region
// calls keep coming.  Block 0 is the DISPATCHER: it polls a ring of 64-byte entries in pinned host memory and, for every
// new submission, writes the region's record into device memory and appends its first ready stage.  Every other block is a
// WORKER: it claims one task at a time from the oldest ready record (one atomic add), runs it -- the same device bodies the
// launched kernels run: prep_read_wave, forward_read<16, K>, post_best / pick_read, sw_align_body<64, K, transposed>,
// project_read -- and counts it in; the worker that completes a stage appends the stages that waited for it, the one that
// completes the last stage stores the finish word into the caller's pinned mirror.  No task ever waits for another task:
// only ready stages are on the list, so a wave is either computing or polling for work, and there is no order in which
// the blocks have to be resident.
//
// Results do not depend on what else is in flight: a submission's forward geometry (16 lanes x fwd_k columns per pair) is
// a function of its own longest haplotype, every task computes its own pairs from the staged inputs, and the integer steps
// are exact -- a region gives the same bits alone, beside nine others, or resubmitted (tests/test_server_hip.py).
//
// Compiled with -ffp-contract=off (the pre-step's threshold and the post-step round like the reference; the forward sweep's
// fused operations are explicit fma calls and inline assembly, which the flag does not touch).
#include "phmm_cigar_device.hpp"
#include "phmm_device.hpp"
#include "phmm_prep_device.hpp"
#include "phmm_server.hpp"
#include "phmm_sw_device.hpp"

namespace phmm {

namespace {

typedef uint32_t srv_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_system(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// What the stages of a call hand each other (staged inputs, modified qualities, the likelihood matrix, best alleles, alignments)
// lies in ordinary device memory, and the eight XCDs' L2s are not coherent with each other: a task begins with an agent-scope
// acquire (its XCD's L2 and its CU's L1 forget what they hold) and ends with an agent-scope release (its stores are written back)
// before it is counted.  Measured (NOTEBOOK 20.2): the pair costs ~20 % of the rate at ten callers; UNCACHED arenas with plain
// waits instead lost tasks (a call in ~10 000 never came back); an ACQUIRE inside the polling loop -- one invalidate per poll and
// idle wave -- kept every L2 of the chip empty and made a region call take 2 ms.
__device__ __forceinline__ void task_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void task_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
// (every memory operation of this wave issued so far has completed)
__device__ __forceinline__ void drain_memory_ops() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// ---- posting tasks -------------------------------------------------------------------------------------------------------
// (the whole wave) tasks [idx0, idx0 + n) of `kind` of region slot `region` go to the next n tickets of class `cls`
__device__ void post_to(const SrvParams &P, uint32_t cls, uint32_t region, uint32_t kind, uint32_t idx0, uint32_t n) {
    if (!n) return;
    const uint32_t lane = threadIdx.x;
    SrvMail *ring = P.mail + (size_t)cls * SRV_MAIL;
    uint32_t first = 0;
    if (lane == 0) first = __hip_atomic_fetch_add(cls ? &P.ctl->posted1 : &P.ctl->posted0, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    first = __builtin_amdgcn_readfirstlane(first);
    for (uint32_t i = lane; i < n; i += WAVE) {
        SrvMail *m = &ring[(first + i) & (SRV_MAIL - 1)];
        st_agent(&m->region, region);
        st_agent(&m->kind, kind);
        st_agent(&m->idx, idx0 + i);
    }
    task_release();  // (the words above are in place before the tags say so)
    for (uint32_t i = lane; i < n; i += WAVE) st_agent(&ring[(first + i) & (SRV_MAIL - 1)].tag, first + i + 1u);
}

// (the whole wave) `n` tasks of `kind` become ready: to the primaries that are waiting for work, then to the secondaries that
// are; what is left when nobody waits is split between the two lines (whoever comes free takes from its own).
__device__ void post(const SrvParams &P, uint32_t region, uint32_t kind, uint32_t n) {
    uint32_t n0 = 0, n1 = 0;
    if (threadIdx.x == 0) {
        const int32_t wait0 = (int32_t)(ld_agent(&P.ctl->next_ticket0) - ld_agent(&P.ctl->posted0));
        const int32_t wait1 = (int32_t)(ld_agent(&P.ctl->next_ticket1) - ld_agent(&P.ctl->posted1));
        n0 = (uint32_t)min(max(wait0, 0), (int32_t)n);
        n1 = (uint32_t)min(max(wait1, 0), (int32_t)(n - n0));
        const uint32_t rest = n - n0 - n1;
        n0 += (rest + 1) / 2;
        n1 += rest / 2;
    }
    n0 = __builtin_amdgcn_readfirstlane(n0);
    n1 = __builtin_amdgcn_readfirstlane(n1);
    post_to(P, 0, region, kind, 0, n0);
    post_to(P, 1, region, kind, n0, n1);
}

// (the whole wave, behind its release) the last task of stage `kind` is through: what waited for it is posted.
//   STAGE -> PREP (-> FWD) and, where the call aligns every pair, SWALL;  FWD (+ SWALL) -> POST;  POST -> SW -> PROJ.
// The last stage stores the finish word for the caller.
__device__ void stage_complete(const SrvParams &P, uint32_t region, SrvRegion *reg, uint32_t kind) {
    const uint32_t lane = threadIdx.x;
    const bool all_pairs = reg->n[SRV_SWALL] != 0;
    bool last = false;
    switch (kind) {
        case SRV_STAGE:
            post(P, region, SRV_PREP, reg->n[SRV_PREP]);
            if (all_pairs) post(P, region, SRV_SWALL, reg->n[SRV_SWALL]);
            break;
        case SRV_PREP: post(P, region, SRV_FWD, reg->n[SRV_FWD]); break;
        case SRV_FWD:
        case SRV_SWALL: {  // (a call that aligns every pair: its post-step waits for both)
            uint32_t both = 1;
            if (all_pairs) {
                if (lane == 0) both = __hip_atomic_fetch_add(&reg->arrived[SRV_POST], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u ? 1u : 0u;
                both = __builtin_amdgcn_readfirstlane(both);
                if (both) task_acquire();
            }
            if (both) post(P, region, SRV_POST, reg->n[SRV_POST]);
            break;
        }
        case SRV_POST:
            if (reg->n[SRV_SW]) post(P, region, SRV_SW, reg->n[SRV_SW]);
            else last = true;
            break;
        case SRV_SW: post(P, region, SRV_PROJ, reg->n[SRV_PROJ]); break;
        default: last = true; break;
    }
    if (last && lane == 0) {
        // (every task's stores are behind its own release and this wave's acquire of the count; now for the host)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(reg->job->finish_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(&P.ctl->finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the tasks ---------------------------------------------------------------------------------------------------------
// (Arguments of a called function arrive in vector registers; every one of these is wave-uniform, and saying so --
// readfirstlane -- is what lets the bodies keep their scalar loads, scalar loop counters and SGPR asm operands.)
template <class T>
__device__ __forceinline__ T *uniform(T *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T *>((uint64_t)hi << 32 | lo);
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// A copy of a job's parameter block with every word in a SCALAR register, as a launched kernel has its arguments (the loads
// themselves are vector loads of one address: the compiler does not issue scalar loads from memory a function might see
// written).
template <class T>
__device__ __forceinline__ T fetch_uniform(const T *from) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    T v;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(uniform(from));
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) w[i] = __builtin_amdgcn_readfirstlane(src[i]);
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}

__device__ __noinline__ void task_stage(const SrvRegion *reg_v, uint32_t idx_v) {
    const SrvRegion *reg = uniform(reg_v);
    const uint32_t idx = uniform(idx_v);
    const srv_u32x4 *src = reinterpret_cast<const srv_u32x4 *>(reg->stage_src);
    srv_u32x4 *dst = reinterpret_cast<srv_u32x4 *>(reg->stage_dst);
    const uint32_t n16 = reg->stage_n16, base = idx * SRV_STAGE_UNITS + threadIdx.x;
    constexpr int ROUNDS = SRV_STAGE_UNITS / WAVE;
    srv_u32x4 v[ROUNDS];
    // (every load of the task in flight over the link before the first store)
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const uint32_t i = base + (uint32_t)j * WAVE;
        if (i < n16) v[j] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const uint32_t i = base + (uint32_t)j * WAVE;
        if (i < n16) dst[i] = v[j];
    }
}

__device__ __noinline__ void task_prep(const SrvJob *job_v, uint32_t idx_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t idx = uniform(idx_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // (named here: a pointer handed in would be a flat one)
    const PrepParams p = fetch_uniform(&job->prep);
    const uint32_t r = idx / p.waves_per_read, c = idx % p.waves_per_read;
    if (r < p.n_reads) prepdev::prep_read_wave(p, r, c, smem);
}

template <int L, int K>
__device__ __noinline__ void task_fwd(const SrvJob *job_v, uint32_t idx_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t idx = uniform(idx_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // (named here: a pointer handed in would be a flat one)
    const ForwardParams p = fetch_uniform(&job->fwd);
    const uint32_t quads = fetch_uniform(&job->fwd_quads);
    const uint32_t r = idx / quads, quad = idx % quads;
    forward_read<L, K>(p, r, (int)quad, (int)quads, false, smem);
}

template <int K>
__device__ __noinline__ void task_sw(const SrvJob *job_v, uint32_t idx_v, uint32_t n_tasks_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t idx = uniform(idx_v), n_tasks = uniform(n_tasks_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // (named here: a pointer handed in would be a flat one)
    const SwParams p = fetch_uniform(&job->sw);
    swdev::sw_align_body<64, K, true>(p, smem, idx, n_tasks, blockIdx.x);
}

// post-step and best allele of up to 64 reads (phmm_post_best_reads' body)
__device__ __noinline__ void task_post(const SrvJob *job_v, uint32_t idx_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t idx = uniform(idx_v);
    const PostBestParams p = fetch_uniform(&job->pb);
    const uint32_t r = idx * WAVE + threadIdx.x;
    if (r >= p.post.n_reads) return;
    const uint32_t g = p.post.read_region[r];
    const uint32_t nh = p.post.region_hap_off[g + 1] - p.post.region_hap_off[g];
    if (nh <= 16) {
        post_best_in_registers<16>(p, r, g, nh);
        return;
    }
    const uint8_t keep = post_read(p.post, r, true);
    if (p.keep_final) p.keep_final[r] = keep;
    best_allele_of(p.best, r, g, keep != 0, !(p.skip_single_allele && nh == 1));
}

// ... of job->proj_per_task reads, whose alignments to every haplotype are there already: the best allele picks its slot
__device__ __noinline__ void task_pick(const SrvJob *job_v, uint32_t idx_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t idx = uniform(idx_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // (named here: a pointer handed in would be a flat one)
    const PostBestParams pb = fetch_uniform(&job->pb);
    const ProjectParams pj = fetch_uniform(&job->pj);
    const uint32_t per_task = fetch_uniform(&job->proj_per_task);
    const uint32_t r = idx * per_task + threadIdx.x;
    if (threadIdx.x < per_task && r < pj.n_reads) cigdev::pick_read(pb, pj, r, reinterpret_cast<uint32_t *>(smem) + (size_t)threadIdx.x * 4 * pj.capacity);
}

__device__ __noinline__ void task_proj(const SrvJob *job_v, uint32_t idx_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t idx = uniform(idx_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // (named here: a pointer handed in would be a flat one)
    const ProjectParams pj = fetch_uniform(&job->pj);
    const uint32_t per_task = fetch_uniform(&job->proj_per_task);
    const uint32_t r = idx * per_task + threadIdx.x;
    if (threadIdx.x < per_task && r < pj.n_reads) cigdev::project_read(pj, r, reinterpret_cast<uint32_t *>(smem) + (size_t)threadIdx.x * 4 * pj.capacity);
}

#define PHMM_SRV_FWD_K(X) \
    X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25)
#define PHMM_SRV_FWD_K32(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#define PHMM_SRV_SW_K(X) X(2) X(3) X(4) X(5) X(6) X(8)

// ---- the dispatcher -------------------------------------------------------------------------------------------------------
__device__ void dispatcher(const SrvParams &P) {
    const uint32_t lane = threadIdx.x;
    uint32_t consumed = P.start_seq;
    uint64_t last_activity = wall_clock64(), last_progress = last_activity;
    uint32_t finished_seen = 0, fault = 0;
    for (;;) {
        // ---- one look at the ring: the entry `consumed` would be in (16 lanes, a dword each) ----------------------------------
        const uint32_t *e = reinterpret_cast<const uint32_t *>(&P.ring[consumed & (SRV_RING - 1)]);
        uint32_t w = lane < 16 ? ld_system(e + lane) : 0u;
        if (__builtin_amdgcn_readfirstlane(w) == consumed + 1u) {
            // (`valid` was stored last; whatever order the link delivered this line's words in, a second look has them all)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            w = lane < 16 ? ld_system(e + lane) : 0u;
            const uint32_t slot_region = consumed & (SRV_RING - 1);
            SrvRegion *reg = &P.regions[slot_region];
            uint32_t total = 0;
#pragma unroll
            for (uint32_t k = 0; k < SRV_KINDS; ++k) {
                const uint32_t nk = __builtin_amdgcn_readlane(w, 2 + k);
                total += nk;
                if (lane == 0) {
                    reg->n[k] = nk;
                    st_agent(&reg->done[k], 0u);
                    st_agent(&reg->arrived[k], 0u);
                }
            }
            if (lane == 0) {
                // (readlane returns a signed int: through uint32_t, or a low half with its top bit set smears over the high half)
                auto word = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane(w, i); };
                const uint32_t flags = word(9), n16 = word(10), job_off = word(11);
                const uint64_t src = (uint64_t)word(12) | (uint64_t)word(13) << 32;
                const uint64_t dst = (uint64_t)word(14) | (uint64_t)word(15) << 32;
                reg->seq = consumed;
                reg->flags = flags;
                reg->stage_n16 = n16;
                reg->stage_src = reinterpret_cast<const void *>(src);
                reg->stage_dst = reinterpret_cast<void *>(dst);
                reg->job = reinterpret_cast<const SrvJob *>(dst + job_off);
            }
            (void)total;
            task_release();
            post(P, slot_region, SRV_STAGE, __builtin_amdgcn_readlane(w, 2 + SRV_STAGE));
            consumed += 1;
            last_activity = last_progress = wall_clock64();
            continue;
        }
        // ---- nothing new: leave when nothing is in flight and nothing has come for a while ------------------------------------
        const uint64_t now = wall_clock64();
        const uint32_t finished = ld_agent(&P.ctl->finished);
        if (finished != finished_seen) {
            finished_seen = finished;
            last_progress = now;
        }
        const uint32_t in_flight = (consumed - P.start_seq) - finished;
        if (in_flight) {
            last_activity = now;
            if (now - last_progress > (uint64_t)P.stall_ticks) {  // (a task that never ends, a stage that never becomes ready)
                fault = 1;
                break;
            }
        } else if (now - last_activity > (uint64_t)P.idle_ticks) {
            break;
        }
    }
    // Everybody out: a worker that takes a ticket from now on sees `closed`; those that are waiting on a ticket get a word each.
    if (lane == 0) {
        if (fault) st_agent(&P.ctl->fault, 1u);
        // (the store has to be out before the loads are issued: relaxed accesses with the wave's memory counter drained in between)
        st_agent(&P.ctl->closed, 1u);
    }
    drain_memory_ops();
    for (uint32_t cls = 0; cls < 2; ++cls) {
        const uint32_t waiting_to = __builtin_amdgcn_readfirstlane(ld_agent(cls ? &P.ctl->next_ticket1 : &P.ctl->next_ticket0));
        const uint32_t waiting_from = __builtin_amdgcn_readfirstlane(ld_agent(cls ? &P.ctl->posted1 : &P.ctl->posted0));
        if ((int32_t)(waiting_to - waiting_from) > 0 && waiting_to - waiting_from < SRV_MAIL)
            for (uint32_t t = waiting_from + lane; (int32_t)(waiting_to - t) > 0; t += WAVE)
                st_agent(&P.mail[(size_t)cls * SRV_MAIL + (t & (SRV_MAIL - 1))].tag, SRV_MAIL_EXIT);
    }
    if (lane == 0) {
        // (the host starts the next launch from `consumed`; that launch runs behind this one on the server's stream)
        P.exit_word->consumed = consumed;
        P.exit_word->fault = fault;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(&P.exit_word->epoch, P.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

__global__ __launch_bounds__(WAVE, 2) void phmm_region_server(const SrvParams P) {
    if (blockIdx.x == 0) {
        dispatcher(P);
        return;
    }
    const uint32_t lane = threadIdx.x;
    // The first of the server's waves on a SIMD is its primary worker, the second its secondary (SrvCtl): where this wave runs,
    // from the hardware's own registers -- HW_ID: SIMD [5:4], CU [11:8], SH [12], SE [15:13]; XCC_ID [3:0].
    uint32_t cls = 0;
    if (lane == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        const uint32_t where = (xcc & 15u) << 9 | ((hw >> 13) & 7u) << 7 | ((hw >> 12) & 1u) << 6 | ((hw >> 8) & 15u) << 2 | ((hw >> 4) & 3u);
        cls = __hip_atomic_fetch_add(&P.ctl->simd_waves[where & 8191u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    }
    cls = __builtin_amdgcn_readfirstlane(cls);
    SrvMail *const ring = P.mail + (size_t)cls * SRV_MAIL;
    uint32_t *const my_tickets = cls ? &P.ctl->next_ticket1 : &P.ctl->next_ticket0;
    for (;;) {
        // ---- take a ticket and wait for its mailbox (lane 0) ------------------------------------------------------------------------
        uint32_t got = 0, region = 0, kind = 0, idx = 0;
        uint64_t t_claim = 0;
        if (lane == 0) {
            const uint32_t ticket = __hip_atomic_fetch_add(my_tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            drain_memory_ops();  // (the ticket is taken before `closed` is looked at)
            SrvMail *m = &ring[ticket & (SRV_MAIL - 1)];
            // (the dispatcher sets `closed`, THEN reads next_ticket and tells every ticket below that to leave: a ticket taken
            // later sees the flag here)
            if (!ld_agent(&P.ctl->closed)) {
                for (uint32_t naps = 0;; ++naps) {
                    // (a RELAXED load: an acquire here is a cache invalidate per poll and wave -- two thousand pollers then keep every
                    // L2 of the chip empty and each task's loads go to memory: 600 us for a PairHMM task of 50.  The one acquire
                    // that is needed follows the claim, below.)
                    const uint32_t tag = ld_agent(&m->tag);
                    if (tag == ticket + 1u) {
                        region = ld_agent(&m->region);
                        kind = ld_agent(&m->kind);
                        idx = ld_agent(&m->idx);
                        got = 1;
                        break;
                    }
                    if (tag == SRV_MAIL_EXIT) break;
                    if ((naps & 1023u) == 1023u && ld_agent(&P.ctl->closed)) break;  // (belt and braces: once in ~1 ms)
                    // nobody else polls this word: ~0.25 us naps while a stage may be about to complete, then ~1 us
                    if (naps < 32) __builtin_amdgcn_s_sleep(8);
                    else __builtin_amdgcn_s_sleep(32);
                }
            }
            if (got && P.trace) t_claim = wall_clock64();
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (!got) return;  // closed
        region = __builtin_amdgcn_readfirstlane(region);
        kind = __builtin_amdgcn_readfirstlane(kind);
        idx = __builtin_amdgcn_readfirstlane(idx);
        task_acquire();  // what the stages before this one stored
        SrvRegion *reg = &P.regions[region];
        const SrvJob *job = reg->job;
        const uint32_t n_kind = reg->n[kind];
        const uint64_t t_begin = P.trace ? wall_clock64() : 0;
        // ---- run it ---------------------------------------------------------------------------------------------------------------
        switch (kind) {
            case SRV_STAGE: task_stage(reg, idx); break;
            case SRV_PREP: task_prep(job, idx); break;
            case SRV_FWD:
                if (job->fwd_l == 32) {
                    switch (job->fwd_k) {
#define PHMM_CASE(KK) \
    case KK: task_fwd<32, KK>(job, idx); break;
                        PHMM_SRV_FWD_K32(PHMM_CASE)
#undef PHMM_CASE
                        default: break;
                    }
                } else {
                    switch (job->fwd_k) {
#define PHMM_CASE(KK) \
    case KK: task_fwd<16, KK>(job, idx); break;
                        PHMM_SRV_FWD_K(PHMM_CASE)
#undef PHMM_CASE
                        default: break;
                    }
                }
                break;
            case SRV_SWALL:
            case SRV_SW:
                switch (job->sw_k) {
#define PHMM_CASE(KK) \
    case KK: task_sw<KK>(job, idx, n_kind); break;
                    PHMM_SRV_SW_K(PHMM_CASE)
#undef PHMM_CASE
                    default: break;
                }
                break;
            case SRV_POST:
                if (job->all_pairs) task_pick(job, idx);
                else task_post(job, idx);
                break;
            case SRV_PROJ: task_proj(job, idx); break;
            default: break;
        }
        // ---- count it in; the task that completes its stage makes the next ones ready --------------------------------------------
        task_release();
        uint32_t completes = 0;
        if (lane == 0) {
            if (P.trace && (reg->flags & 1u)) {
                const uint32_t t = __hip_atomic_fetch_add(&P.ctl->trace_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t < P.trace_cap) {
                    SrvTrace &tr = P.trace[t];
                    tr.seq = reg->seq;
                    tr.kind = kind;
                    tr.idx = idx;
                    tr.worker = blockIdx.x | cls << 31;
                    tr.t_claim = t_claim;
                    tr.t_begin = t_begin;
                    tr.t_end = wall_clock64();
                }
            }
            completes = __hip_atomic_fetch_add(&reg->done[kind], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == n_kind ? 1u : 0u;
        }
        if (__builtin_amdgcn_readfirstlane(completes)) {
            task_acquire();
            stage_complete(P, region, reg, kind);
        }
    }
}

hipError_t launch_server(const SrvParams &p, uint32_t n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(phmm_region_server, dim3(n_blocks), dim3(WAVE), SRV_LDS_BYTES, stream, p);
    return hipGetLastError();
}

int server_blocks_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, phmm_region_server, WAVE, SRV_LDS_BYTES) != hipSuccess) nb = 0;
    return nb;
}

}  // namespace phmm
