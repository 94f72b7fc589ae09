// gfx950: the resident region server (phmm_server.hpp) -- one kernel of one-wave blocks that stays on the chip while region
// calls keep coming.  Block 0 is the DISPATCHER: it polls a ring of 64-byte entries in pinned host memory and, for every new
// submission, writes the region's record into device memory and posts its stage-in tasks.  Every other block is a WORKER: it
// takes a ticket, waits for the mailbox of that ticket (a word nobody else polls), runs the task and counts it in.  Two kinds
// of task: STAGE copies 16 KB of the call's inputs from the pinned mirror; CHAIN is one READ of the call from start to end
// in one wave -- prep_read_wave, forward_read<16, K> against a group of four haplotypes (helper waves take the further groups
// and hand their four likelihoods over), post_best, sw_align_body<64, K, transposed> against the best haplotype, project_read:
// the same device bodies the launched kernels run.  A read's intermediate results never leave its wave, so nothing waits for a
// stage to complete and nothing has to be made coherent between the chip's eight L2s except the staged inputs (once per call
// and XCD) and a helper's four numbers.
//
// Results do not depend on what else is in flight: a submission's forward geometry (16 lanes x fwd_k columns per pair) is a
// function of its own longest haplotype, every wave computes its own pairs from the staged inputs, and the integer steps are
// exact -- a region gives the same bits beside one other caller or thirty, or resubmitted (tests/test_server_hip.py).
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
template <class T>
__device__ __forceinline__ T *ld_agent_ptr(T *const *p) {
    return reinterpret_cast<T *>(__hip_atomic_load(reinterpret_cast<const uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// ---- coherence ----------------------------------------------------------------------------------------------------------------
// The eight XCDs' L2s are not coherent with each other.  Cache-wide operations are what made a first, staged version of this
// server slow (every task began with an invalidate and ended with a write-back of its XCD's whole L2: 1 300 of each per region,
// tasks twice as long as alone; and an ACQUIRE inside the polling loop -- an invalidate per poll and idle wave -- made a region
// call take 2 ms).  Here:
//   * control words (tickets, mailboxes, counters, region records, a helper's likelihoods) are accessed with agent-scope
//     atomics only, which are coherent where they are;
//   * a task ends with one agent-scope release: nothing of it stays dirty in an L2, so a slot's next call (any XCD) finds memory
//     as this call left it and no stale line can be written over newer data later;
//   * the staged inputs are the only bulk data one wave writes and others read: a chain task begins with one agent-scope acquire.
// Two cache-wide operations per read and group of haplotypes, where the staged version had ten per read.
__device__ __forceinline__ void task_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
// (every memory operation of this wave issued so far has completed)
__device__ __forceinline__ void drain_memory_ops() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
// (the whole wave) what the call's stage-in wrote is what this wave reads from now on: its XCD's L2 and its CU's L1 forget what
// they hold.  (Measured and dropped: emptying the L2 only once per XCD and stage-in, with an L1-only invalidate -- buffer_inv sc0
// -- for the other tasks: reads of stale inputs, one call in a few hundred wrong; and no faster.)
__device__ __forceinline__ void acquire_inputs() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

// ---- posting tasks -------------------------------------------------------------------------------------------------------
// (the whole wave) `n` tasks of `kind` of region slot `region` go to the next n tickets, a mailbox each (phmm_server.hpp)
__device__ void post(const SrvParams &P, uint32_t region, uint32_t kind, uint32_t n) {
    if (!n) return;
    const uint32_t lane = threadIdx.x;
    uint32_t first = 0;
    if (lane == 0) first = __hip_atomic_fetch_add(&P.ctl->posted, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    first = __builtin_amdgcn_readfirstlane(first);
    for (uint32_t i = lane; i < n; i += WAVE) {
        SrvMail *m = &P.mail[(first + i) & (SRV_MAIL - 1)];
        st_agent(&m->region, region);
        st_agent(&m->kind, kind);
        st_agent(&m->idx, i);
    }
    drain_memory_ops();  // (the words above are in place before the tags say so)
    for (uint32_t i = lane; i < n; i += WAVE) st_agent(&P.mail[(first + i) & (SRV_MAIL - 1)].tag, first + i + 1u);
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

__device__ __noinline__ void task_stage(const void *src_v, void *dst_v, uint32_t n16_v, uint32_t idx_v) {
    const srv_u32x4 *src = reinterpret_cast<const srv_u32x4 *>(uniform(src_v));
    srv_u32x4 *dst = reinterpret_cast<srv_u32x4 *>(uniform(dst_v));
    const uint32_t n16 = uniform(n16_v), base = uniform(idx_v) * SRV_STAGE_UNITS + threadIdx.x;
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

// the read's pre-step: every wave of the read runs it (the same values to the same places) and reads back what it wrote
__device__ __noinline__ void chain_prep(const SrvJob *job_v, uint32_t r_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t r = uniform(r_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // (named here: a pointer handed in would be a flat one)
    const PrepParams p = fetch_uniform(&job->prep);
    prepdev::prep_read_wave(p, r, 0, smem);
}

template <int L, int K>
__device__ __noinline__ void chain_fwd(const SrvJob *job_v, uint32_t r_v, uint32_t group_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t r = uniform(r_v), group = uniform(group_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ForwardParams p = fetch_uniform(&job->fwd);
    const uint32_t groups = fetch_uniform(&job->groups);
    double *const helper_row = group ? fetch_uniform(&job->helper_out) + (size_t)r * fetch_uniform(&job->helper_stride) : nullptr;
    forward_read<L, K>(p, r, (int)group, (int)groups, false, smem, helper_row);
}

// (lane 0) the helpers' likelihoods into this wave's own view of the row, then post-step and best allele of the read
__device__ __noinline__ void chain_post(const SrvJob *job_v, uint32_t r_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t r = uniform(r_v);
    if (threadIdx.x != 0) return;
    const PostBestParams p = fetch_uniform(&job->pb);
    const uint32_t g = p.post.read_region[r];
    const uint32_t nh = p.post.region_hap_off[g + 1] - p.post.region_hap_off[g];
    double *row = p.post.out + p.post.out_off[g] + (uint64_t)(r - p.post.region_read_off[g]) * nh;
    const double *helper_row = fetch_uniform(&job->helper_out) + (size_t)r * fetch_uniform(&job->helper_stride);
    for (uint32_t a = fetch_uniform(&job->group_haps); a < nh; ++a)  // (groups 1.. are the helpers'; stored with agent-scope stores, read the same way)
        row[a] = __hip_atomic_load(&helper_row[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nh <= 16) {
        post_best_in_registers<16>(p, r, g, nh);
        return;
    }
    const uint8_t keep = post_read(p.post, r, true);
    if (p.keep_final) p.keep_final[r] = keep;
    best_allele_of(p.best, r, g, keep != 0, !(p.skip_single_allele && nh == 1));
}

template <int K>
__device__ __noinline__ void chain_sw(const SrvJob *job_v, uint32_t r_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t r = uniform(r_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SwParams p = fetch_uniform(&job->sw);
    swdev::sw_align_body<64, K, true>(p, smem, r, p.n_alignments, blockIdx.x);  // (alignment r and no other)
}

__device__ __noinline__ void chain_proj(const SrvJob *job_v, uint32_t r_v) {
    const SrvJob *job = uniform(job_v);
    const uint32_t r = uniform(r_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x != 0) return;
    const ProjectParams pj = fetch_uniform(&job->pj);
    cigdev::project_read(pj, r, reinterpret_cast<uint32_t *>(smem));
}

#define PHMM_SRV_FWD_K(X) \
    X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25)
#define PHMM_SRV_FWD_K32(X) X(13) X(14) X(15) X(16)
#define PHMM_SRV_SW_K(X) X(2) X(3) X(4) X(5) X(6) X(8)

// ---- the dispatcher -------------------------------------------------------------------------------------------------------
__device__ void dispatcher(const SrvParams &P) {
    const uint32_t lane = threadIdx.x;
    uint32_t consumed = P.start_seq;
    uint64_t last_activity = wall_clock64(), last_progress = last_activity;
    uint32_t finished_seen = 0, fault = 0;
    const uint32_t yield_at_start = __builtin_amdgcn_readfirstlane(ld_system(P.yield_word));
    bool leaving = false;  // the host wants the chip for launched kernels: no further call is taken
    for (;;) {
        // ---- one look at the ring: the entry `consumed` would be in (16 lanes, a dword each; lane 16: the yield word) ----------
        const uint32_t *e = reinterpret_cast<const uint32_t *>(&P.ring[consumed & (SRV_RING - 1)]);
        uint32_t w = lane < 16 ? ld_system(e + lane) : lane == 16 ? ld_system(P.yield_word) : 0u;
        if ((uint32_t)__builtin_amdgcn_readlane(w, 16) != yield_at_start) leaving = true;
        if (!leaving && __builtin_amdgcn_readfirstlane(w) == consumed + 1u) {
            // (`valid` was stored last; whatever order the link delivered this line's words in, a second look has them all)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            w = lane < 16 ? ld_system(e + lane) : 0u;
            // (readlane returns a signed int: through uint32_t, or a low half with its top bit set smears over the high half)
            auto word = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane(w, i); };
            const uint32_t slot_region = consumed & (SRV_RING - 1);
            SrvRegion *reg = &P.regions[slot_region];
            if (lane == 0) {
                const uint64_t src = (uint64_t)word(12) | (uint64_t)word(13) << 32;
                const uint64_t dst = (uint64_t)word(14) | (uint64_t)word(15) << 32;
                st_agent(&reg->seq, consumed);
                st_agent(&reg->n[SRV_STAGE], word(2));
                st_agent(&reg->n[SRV_CHAIN], word(3));
                st_agent(&reg->flags, word(4));
                st_agent(&reg->stage_n16, word(5));
                st_agent(&reg->done[SRV_STAGE], 0u);
                st_agent(&reg->done[SRV_CHAIN], 0u);
                st_agent(&reg->timed_out, 0u);
                __hip_atomic_store(reinterpret_cast<uint64_t *>(&reg->stage_src), src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<uint64_t *>(&reg->stage_dst), dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<uint64_t *>(&reg->job), dst + word(6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            drain_memory_ops();
            post(P, slot_region, SRV_STAGE, word(2));
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
            if (now - last_progress > (uint64_t)P.stall_ticks) {  // (a task that never ends)
                fault = 1;
                break;
            }
        } else if (leaving || now - last_activity > (uint64_t)P.idle_ticks) {
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
    {
        const uint32_t waiting_to = __builtin_amdgcn_readfirstlane(ld_agent(&P.ctl->next_ticket));
        const uint32_t waiting_from = __builtin_amdgcn_readfirstlane(ld_agent(&P.ctl->posted));
        if ((int32_t)(waiting_to - waiting_from) > 0 && waiting_to - waiting_from < SRV_MAIL)
            for (uint32_t t = waiting_from + lane; (int32_t)(waiting_to - t) > 0; t += WAVE) st_agent(&P.mail[t & (SRV_MAIL - 1)].tag, SRV_MAIL_EXIT);
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
    for (;;) {
        // ---- take a ticket and wait for its mailbox (lane 0) ------------------------------------------------------------------------
        uint32_t got = 0, region = 0, kind = 0, idx = 0;
        uint64_t t_claim = 0;
        if (lane == 0) {
            const uint32_t ticket = __hip_atomic_fetch_add(&P.ctl->next_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            drain_memory_ops();  // (the ticket is taken before `closed` is looked at)
            SrvMail *m = &P.mail[ticket & (SRV_MAIL - 1)];
            // (the dispatcher sets `closed`, THEN reads next_ticket and tells every ticket below that to leave: a ticket taken
            // later sees the flag here)
            if (!ld_agent(&P.ctl->closed)) {
                for (uint32_t naps = 0;; ++naps) {
                    // (a RELAXED load: an acquire here is a cache invalidate per poll and wave -- two thousand pollers then keep every
                    // L2 of the chip empty and each task's loads go to memory)
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
                    // nobody else polls this word: ~0.25 us naps at first, then ~1 us
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
        SrvRegion *reg = &P.regions[region];
        // (the region record by agent-scope loads: this wave's caches may hold what the ring slot's previous call left)
        const uint32_t n_kind = __builtin_amdgcn_readfirstlane(ld_agent(&reg->n[kind]));
        const uint32_t reg_flags = __builtin_amdgcn_readfirstlane(ld_agent(&reg->flags));
        const SrvJob *job = uniform(ld_agent_ptr(&reg->job));
        const uint64_t t_begin = P.trace ? wall_clock64() : 0;
        uint64_t t_mid[4] = {0, 0, 0, 0};
        // ---- run it ---------------------------------------------------------------------------------------------------------------
        if (kind == SRV_STAGE) {
            task_stage(ld_agent_ptr(&reg->stage_src), ld_agent_ptr(&reg->stage_dst), ld_agent(&reg->stage_n16), idx);
        } else {
            acquire_inputs();
            const uint32_t groups = uniform(job->groups), r = idx / groups, group = groups - 1u - idx % groups;  // (helpers first, the main wave last)
            chain_prep(job, r);
            drain_memory_ops();  // (what the pre-step stored is what the sweep's row staging loads)
            if (P.trace) t_mid[0] = wall_clock64();
            if (uniform(job->group_haps) == 2u) {  // (haplotypes beyond 400 bases: 32 lanes per pair, two a wave)
                switch (uniform(job->fwd_k)) {
#define PHMM_CASE(KK) \
    case KK: chain_fwd<32, KK>(job, r, group); break;
                    PHMM_SRV_FWD_K32(PHMM_CASE)
#undef PHMM_CASE
                    default: break;
                }
            } else {
                switch (uniform(job->fwd_k)) {
#define PHMM_CASE(KK) \
    case KK: chain_fwd<16, KK>(job, r, group); break;
                    PHMM_SRV_FWD_K(PHMM_CASE)
#undef PHMM_CASE
                    default: break;
                }
            }
            if (P.trace) t_mid[1] = wall_clock64();
            uint32_t *group_done = uniform(job->group_done) + r;
            if (group != 0) {  // a helper: its likelihoods are out (agent-scope stores), the main wave may count on them
                drain_memory_ops();
                if (lane == 0) __hip_atomic_fetch_add(group_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                uint32_t in_time = 1;
                if (groups > 1 && lane == 0) {  // (the helpers hold earlier tickets than this wave: they are running or through)
                    const uint64_t t0 = wall_clock64();
                    while (ld_agent(group_done) != groups - 1u) {
                        if (wall_clock64() - t0 > (uint64_t)job->wait_ticks) {
                            in_time = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                if (__builtin_amdgcn_readfirstlane(in_time)) {
                    chain_post(job, r);
                    drain_memory_ops();  // (the best haplotype is where the aligner looks it up)
                    if (P.trace) t_mid[2] = wall_clock64();
                    switch (uniform(job->sw_k)) {
#define PHMM_CASE(KK) \
    case KK: chain_sw<KK>(job, r); break;
                        PHMM_SRV_SW_K(PHMM_CASE)
#undef PHMM_CASE
                        default: break;
                    }
                    drain_memory_ops();  // (... and the alignment where the projection reads it)
                    if (P.trace) t_mid[3] = wall_clock64();
                    chain_proj(job, r);
                } else if (lane == 0) {
                    st_agent(&reg->timed_out, 1u);
                }
            }
        }
        // ---- count it in: nothing of the task stays dirty in this XCD's L2; the last chain task of a call tells the caller ----------
        task_release();
        uint32_t completes = 0;
        if (lane == 0) {
            if (P.trace && (reg_flags & 1u)) {
                const uint32_t t = __hip_atomic_fetch_add(&P.ctl->trace_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t < P.trace_cap) {
                    SrvTrace &tr = P.trace[t];
                    tr.seq = ld_agent(&reg->seq);
                    tr.kind = kind;
                    tr.idx = idx;
                    tr.worker = blockIdx.x;
                    tr.t_claim = t_claim;
                    tr.t_begin = t_begin;
                    tr.t_end = wall_clock64();
                    for (int i = 0; i < 4; ++i) tr.t_mid[i] = t_mid[i];
                }
            }
            completes = __hip_atomic_fetch_add(&reg->done[kind], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == n_kind ? 1u : 0u;
        }
        if (!__builtin_amdgcn_readfirstlane(completes)) continue;
        if (kind == SRV_STAGE) {
            // the inputs are in memory (every stage task's release came before its count): zero the words the call's waves hand each
            // other -- by the same kind of store they are read with --, then post the reads
            acquire_inputs();  // (the job record: this wave staged only a part of it)
            {
                uint32_t *gd = uniform(job->group_done);
                const uint32_t nr = uniform(job->n_reads);
                for (uint32_t i = lane; i < nr; i += WAVE) st_agent(gd + i, 0u);
                if (lane == 0) st_agent(uniform(job->status_in), 0u);
            }
            drain_memory_ops();
            post(P, region, SRV_CHAIN, __builtin_amdgcn_readfirstlane(ld_agent(&reg->n[SRV_CHAIN])));
        } else if (lane == 0) {
            // (every task's stores are behind its own release; now for the host: the sweeps' status word, then the finish word)
            uint32_t status = __hip_atomic_load(job->status_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *job->status_out = status;
            if (ld_agent(&reg->timed_out)) job->status_out[33] = 1u;  // (ProjectParams::flags[1]: a wait inside the call ran out of time)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(job->finish_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_add(&P.ctl->finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
