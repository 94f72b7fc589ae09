// Device-side body of the Smith-Waterman aligner (phmm_sw_align_kernel, phmm_sw_kernels.hip; the resident region server's
// aligner tasks, phmm_server_kernels.hip).  See phmm_sw_kernels.hip for the mapping.
#pragma once
#include <algorithm>
#include <type_traits>
#include "phmm_sw_internal.hpp"

namespace phmm {

namespace swdev {


constexpr int32_t SW_LOW_INIT = INT32_MIN / 2;        // :137 (below every scaled score; its low two bits are clear)
enum : int32_t { TAG_DOWN = 0, TAG_RIGHT = 1, TAG_DIAG = 2 };
enum : uint32_t { OP_M = 0, OP_I = 1, OP_D = 2, OP_S = 4 };
enum : int { ST_MATCH = 0, ST_INSERTION = 1, ST_DELETION = 2, ST_CLIP = 3 };

__device__ __forceinline__ uint32_t make_element(int state, uint32_t length) {  // :445-452
    const uint32_t op = state == ST_MATCH ? OP_M : state == ST_INSERTION ? OP_I : state == ST_DELETION ? OP_D : OP_S;
    return (length << 4) | op;
}

// CIGAR under construction: elements arrive last-to-first (the reference pushes them and reverses at the end).
struct CigarOut {
    uint32_t *slot;
    uint64_t cap;
    bool writer;  // the lanes of an alignment all keep count, one of them writes
    uint32_t n = 0;
    __device__ void push(uint32_t e) {
        if (writer && n < cap) slot[n] = e;
        ++n;
    }
    __device__ void finish() {  // lce.reverse() (:441)
        if (n <= cap)
            for (uint32_t a = 0, b = n ? n - 1 : 0; a < b; ++a, --b) {
                const uint32_t t = slot[a];
                slot[a] = slot[b];
                slot[b] = t;
            }
    }
};



// f(integral_constant<0>), f(<4>), f(<8>) ... while below K
template <int K, int K0 = 0, class F>
__device__ __forceinline__ void static_for_chunks4(F &&f) {
    if constexpr (K0 < K) {
        f(std::integral_constant<int, K0>{});
        static_for_chunks4<K, K0 + 4>(f);
    }
}

template <int SW_L>
__device__ __forceinline__ int32_t row_shr1(int32_t v) {  // lane l <- lane l-1 (first lane of a row of 16 / of the wave: 0)
    if constexpr (SW_L <= 16) return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    else return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);                       // wave_shr:1
}

// Candidate start cells of the backtrack compare as the reference's scans do (:303-330): higher score first; among equal
// scores the smaller |p1 - p2|; among those the one met first (last column before bottom row, bottom row left to right).
struct Start {
    int32_t score, dist, order, p1, p2;
    uint32_t g = 0;  // (the tags-only sweep) the walk from this cell is one diagonal
};
__device__ __forceinline__ bool better(const Start &a, const Start &b) {  // a beats b
    if (a.score != b.score) return a.score > b.score;
    if (a.dist != b.dist) return a.dist < b.dist;
    return a.order < b.order;
}

// a backtrack flag word, read where the stores of this wave went (L2)
__device__ __forceinline__ uint32_t flag_load(const uint32_t *at) {
    return __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NW dwords of backtrack flags, one streaming store (global_store_dword / x2 / x3 / x4; four-byte alignment is all they need)
template <int NW, typename V>
__device__ __forceinline__ void store_flags(uint32_t *at, const V &v) {
    if constexpr (NW == 1) {
        const uint32_t one = v[0];
        asm volatile("global_store_dword %0, %1, off nt" ::"v"(at), "v"(one) : "memory");
    }
    else if constexpr (NW == 2) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(at), "v"(v) : "memory");
    // (a store of more than 64 bits needs two wait states before a VALU instruction may overwrite its data registers on
    // gfx940+, and the compiler's hazard recogniser does not look into inline assembly: the s_nop keeps that distance)
    else if constexpr (NW == 3) asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(at), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(at), "v"(v) : "memory");
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the flag stores ('nt' modifier, hazard distance) and the raw s_waitcnt immediate of the walk are written for gfx942 / gfx950"
#endif

#ifndef PHMM_SW_K4
#define PHMM_SW_K4 19
#endif
#ifndef PHMM_SW_EU
#define PHMM_SW_EU 5
#endif
// SW_L lanes per alignment (8 / 16 / 32 / 64: 8 ... 1 alignments per wave), K columns per lane
// Registers: the allocator's own choice.  An occupancy target (amdgpu_waves_per_eu: 4 waves per SIMD up to K = 19, 5 up to
// K = 12) was worth 2-4 % on the 8 x 19 instance and cost 20-40 spilled VGPRs and 90-150 bytes of scratch per lane in the
// set-up and backtrack code of a dozen instances (PHMM_SW_FORCE_OCCUPANCY builds it back in for A/B runs); since the sweep
// updates the row above in place (one register set instead of two) no instance needs scratch.
#ifdef PHMM_SW_FORCE_OCCUPANCY
#define PHMM_SW_OCCUPANCY(K) __attribute__((amdgpu_waves_per_eu(K <= 12 ? PHMM_SW_EU : K <= PHMM_SW_K4 ? 4 : K <= 26 ? 3 : 2)))
#else
#define PHMM_SW_OCCUPANCY(K)
#endif
// WIDE: weights so large that four times a score no longer fits 32 bits, or that the reference's clamp at -1e8
// (MATRIX_MIN_CUTOFF, :31) can act: scores are carried as they are, the winning candidate is found by the reference's own
// comparisons (:250-266) and the clamp is applied -- six more instructions per cell, one instance (16 lanes x 16 columns).
// EXT: sequences of tens of thousands of bases -- the per-row / per-column arrays (bottom row, strip edges) live in device
// memory, one slice per block, LDS holds the two sequences only.  An instance of its own (16 lanes x 32 columns, and the
// wide one): loads from device memory inside the sweep make the compiler wait for ALL outstanding memory operations of a
// step -- the flag stores included -- which cost the ordinary instances 5-8 % while the two shared one body.
// LITE: the first of two passes where gaps are rare (the usual read against its haplotype) -- 11 instructions per cell instead
// of 16 and, since round 5, NO backtrack flags at all.  A walk that meets no gap is the diagonal from its start cell to the
// matrix's edge, and whether it meets one is a bit the sweep can carry: G(i, j) = [the diagonal candidate won at (i, j)] and
// G(i-1, j-1), G = 1 on row 0 and column 0.  The lane keeps the G bits of its K columns in the same two-bits-per-cell layout the
// tags are collected in (the tag's high bit IS "diagonal won"), so one step costs one shift-or and one AND per 16 columns plus
// the hand-over of one bit to the next lane -- nothing per cell.  The start cell's G decides: 1 = the CIGAR is written from the
// start cell alone (no flag was stored, none is read: 2.6 GB of flags per 131 072 reads and a dozen dependent round trips per
// walk are gone); 0 = the alignment is put on a list and aligned again by the full instance, launched behind this one over that
// list (SwParams::todo), so results never depend on which of the two ran.  One strip only (an alignment with more goes on the list).
// `vblock` of `vgrid`: which share of the alignments this wave takes (a launch: its block of the grid; the resident region
// server: the task's index among the region's aligner tasks); `slab_block`: whose slab of backtrack flags / slice of p.ext
// (a launch: its block; the server: the worker wave's own, whatever task it runs).
template <int SW_L, int K, bool TR = false, bool WIDE = false, bool EXT = false, bool LITE = false>
__device__ __forceinline__ void sw_align_body(const SwParams &p, unsigned char *smem, const uint32_t vblock, const uint32_t vgrid,
                                              const uint32_t slab_block) {
    static_assert(!LITE || (!WIDE && !EXT), "the tags-only sweep exists for the ordinary instances");
    // TR: the sweep runs along the ALTERNATE sequence and the lanes share out the reference's rows (K rows per lane) --
    // the same cells in another order.  For a small call of reads against longer haplotypes that is fewer steps of more
    // cells each (150 x 300 on 64 lanes: 210 steps of five cells instead of 350 of three), and a step's fixed cost is
    // what a lone wave per SIMD feels.  One strip only (the host sees to it).
    constexpr int GMASK = WAVE - SW_L;  // lane & GMASK = first lane of the lane's group
    constexpr uint64_t LMASK = SW_L == 64 ? ~0ull : (1ull << (SW_L & 63)) - 1;  // the group's lanes, shifted down
    const int lane = threadIdx.x, g = lane / SW_L, l = lane % SW_L;
    // LDS of this group: reference | alternate | bottom row | (several strips only) strip edge: sw, best_gap_h, -gap_size_h
    const uint32_t gpb = p.groups_per_block;  // 64 / SW_L, or 1 when the sequences are so long that a block's LDS holds one alignment
    unsigned char *gbase = smem + (size_t)(g < (int)gpb ? g : 0) * p.lds_group_bytes;
    uint8_t *s_ref = gbase;
    uint8_t *s_alt = s_ref + p.lds_ref_bytes;
    int32_t *bottom;
    if constexpr (EXT) bottom = reinterpret_cast<int32_t *>(p.ext + (size_t)slab_block * p.ext_stride);
    else bottom = reinterpret_cast<int32_t *>(s_alt + p.lds_alt_bytes);
    int32_t *e_sw = bottom + (p.max_alt + 1);
    int32_t *e_bgh = e_sw + (p.max_ref + 1);
    // backtrack flags of this block, [strip][step][dword][lane]
    constexpr int NH = (K + 15) / 16;  // flag accumulator pairs per lane: candidate tags, gap-open bits of 16 cells each
    constexpr int NW = LITE ? sw_tag_words(K) : sw_flag_words(K);  // dwords stored per lane and step
    constexpr int REM = K % 16;
    constexpr bool LAST_PACKED = !LITE && REM != 0 && REM <= 8;  // the last pair shares a dword: tags in the top 2 REM bits, gap bits in the bottom 2 REM
    // GBIT: the sweep carries "the walk from this cell is one diagonal" per column (see LITE above).  The tags-only sweep lives on
    // it; the full instance uses it to skip the walk -- a dozen dependent round trips to the flags -- for the alignments that have
    // no gap (SoftClip / Ignore, one strip), which is what a small region call's aligner spends its last ~15 us on.
    constexpr bool GBIT = !WIDE && !EXT;
    // a tag word holds its cells top-aligned: column q of a word of nq cells has its tag's high bit at 33 - 2 (nq - q)
    constexpr int NQ0 = K < 16 ? K : 16, G_IN0 = 33 - 2 * NQ0;   // ... so the first column of the lane sits at G_IN0
    uint32_t *slab = p.slab + (size_t)slab_block * p.slab_stride;
    // scores times four; the low two bits name the candidate
    constexpr int32_t SC = WIDE ? 1 : 4, TG = WIDE ? 0 : 1;  // scale of the scores; whether their low two bits carry the candidate
    // (readfirstlane: nothing for a launched kernel, whose arguments are scalar; the region server's tasks are called functions)
    int32_t x_match = __builtin_amdgcn_readfirstlane(SC * p.w_match + TG * TAG_DIAG), x_mismatch = __builtin_amdgcn_readfirstlane(SC * p.w_mismatch + TG * TAG_DIAG);
    asm volatile("" : "+s"(x_match), "+s"(x_mismatch));  // opaque: or the compiler selects between the raw weights and scales per cell
    // the gap that runs along the sweep is kept per lane position in registers, the one across it travels through the
    // step and on to the next lane: vertical (tag: down) and horizontal (tag: right), or the other way round
    constexpr int32_t TAG_S = TR ? TAG_RIGHT : TAG_DOWN, TAG_L = TR ? TAG_DOWN : TAG_RIGHT;
    const int32_t x_open = SC * p.w_open, x_open_s = SC * p.w_open + TG * TAG_S, x_open_l = SC * p.w_open + TG * TAG_L, x_extend = SC * p.w_extend;
    const bool edge_gaps = p.strategy == PHMM_SW_STRATEGY_INDEL || p.strategy == PHMM_SW_STRATEGY_LEADING_INDEL;  // :145
    const int strip_cols = SW_L * K;
    const size_t strip_stride = (size_t)(p.max_ref + SW_L) * NW * WAVE;  // flag dwords of one strip
    auto row0 = [&](int jj) { return (edge_gaps && jj > 0) ? x_open + (jj - 1) * x_extend : 0; };  // :150-158

    // (block 0 reports the shader clock it ran at -- phmm_get_stat "sw_clock_mhz" -- only where the launch asks for it:
    // SwParams::report_clock, measurement runs of the aligner's own entry points; never inside a region call)
    const long long clk0 = p.report_clock ? clock64() : 0, wall0 = p.report_clock ? wall_clock64() : 0;
    // the alignments of this launch: [a_begin, n_alignments), or the list an earlier tags-only launch left (todo)
    const uint32_t n_items = p.todo ? *p.todo_count : p.n_alignments;
    if (p.todo && (n_items < p.todo_min || (p.todo_max && n_items > p.todo_max))) return;  // (the other geometry's launch takes this list)
    if (p.todo && p.feedback && vblock == 0 && lane == 0) *p.feedback = n_items;  // (the host looks at it between calls)
    for (uint32_t base = (p.todo ? 0u : p.a_begin) + vblock * gpb; base < n_items; base += vgrid * gpb) {
        const uint32_t item = base + (uint32_t)g;
        bool valid = (uint32_t)g < gpb && item < n_items;
        const uint32_t a = !valid ? 0u : p.todo ? p.todo[item] : item;
        uint32_t ro = 0, ao = 0, aa = a;  // aa: the alternate sequence of alignment a
        int n = 0, m = 0;
        if (valid) {
            uint32_t ri;
            if (p.pair_stride) {  // every read against EVERY haplotype of its region: alignment a = (read a / stride, haplotype a % stride)
                aa = a / p.pair_stride;
                const uint32_t j = a - aa * p.pair_stride;
                if (p.pair_single_nh) {  // one region: no look-ups (over PCIe each is a round trip of its own in front of the bases)
                    ri = j < p.pair_single_nh ? j : SW_NO_REFERENCE;
                } else {
                    const uint32_t reg = p.read_region[aa], h0 = p.region_hap_off[reg];
                    ri = j < p.region_hap_off[reg + 1] - h0 ? h0 + j : SW_NO_REFERENCE;
                }
            } else {
                ri = p.ref_index ? p.ref_index[a] : a;  // reads name their haplotype; pairs come one to one
            }
            if (ri == SW_NO_REFERENCE) {  // nothing to align (evidence removed / no allele): an empty CIGAR
                if (l == 0) {
                    p.n_cigar[a] = 0;
                    p.alignment_offset[a] = 0;
                }
                valid = false;
            } else {
                ro = p.ref_off[ri];
                ao = p.alt_off[aa];
                n = (int)(p.ref_off[ri + 1] - ro);
                m = (int)(p.alt_off[aa + 1] - ao);
                if (p.alt_clip) {  // the read minus its soft clips (alignment_utils.rs:47-50)
                    const uint32_t cl = p.alt_clip[2 * aa], cr = p.alt_clip[2 * aa + 1];
                    ao += cl;
                    m -= (int)(cl + cr);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // (one loop: the loads of both sequences are in flight together -- one round trip to memory instead of two)
        for (int k = l, nm = max(n, m); k < nm; k += SW_L) {
            const uint8_t rb = k < n ? p.ref_bases[ro + k] : (uint8_t)0, ab = k < m ? p.alt_bases[ao + k] : (uint8_t)0;
            if (k < n) s_ref[k] = rb;
            if (k < m) s_alt[k] = ab;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- exact substring: SoftClip / Ignore only (:72-81), the LAST occurrence (alignment_utils.rs:717-735) ----
        // Every lane of the group screens one candidate offset by its first eight bases (two dword compares; a random offset
        // passes once in 65 536); what passes -- normally only the offset the read really comes from -- is verified by the
        // whole group together, every lane a share of the dwords.  (One lane comparing byte after byte held its wave for the
        // length of the read at every true offset: an eighth of the kernel's time.)
        int found = -1;
        if (p.strategy == PHMM_SW_STRATEGY_SOFTCLIP || p.strategy == PHMM_SW_STRATEGY_IGNORE) {
            const uint32_t *ref_w = reinterpret_cast<const uint32_t *>(s_ref), *alt_w = reinterpret_cast<const uint32_t *>(s_alt);
            // do alt[q .. q + 4) and ref[r + q .. r + q + 4) differ (bytes from m on do not count)?  q is a multiple of four
            auto differ = [&](int r, int q) {
                const int at = r + q;
                const uint32_t x = __builtin_amdgcn_alignbyte(ref_w[(at >> 2) + 1], ref_w[at >> 2], (uint32_t)at & 3u) ^ alt_w[q >> 2];
                const int left_over = m - q;  // >= 1
                return (left_over >= 4 ? x : x & ((1u << (8 * left_over)) - 1u)) != 0u;
            };
            int r0 = valid ? n - m : -1;
            while (__any(found < 0 && r0 >= 0)) {
                const int r = r0 - l;
                bool cand = found < 0 && r0 >= 0 && r >= 0;
                if (cand) cand = !differ(r, 0) && (m <= 4 || !differ(r, 4));
                uint64_t cmask = (__ballot(cand) >> (lane & GMASK)) & LMASK;  // the group's candidates, highest offset in the lowest bit
                while (__any(cmask != 0ull && found < 0)) {
                    const bool on = cmask != 0ull && found < 0;
                    const int rc = r0 - (on ? __ffsll((long long)cmask) - 1 : 0);
                    bool bad = false;
                    if (on)
                        for (int q = 8 + 4 * l; q < m; q += 4 * SW_L) bad |= differ(rc, q);
                    const bool any_bad = ((__ballot(bad) >> (lane & GMASK)) & LMASK) != 0ull;
                    if (on && !any_bad) found = rc;
                    cmask &= cmask - 1ull;  // next candidate
                }
                r0 -= SW_L;
            }
        }
        const bool dp = valid && found < 0;  // this group runs the matrix

        // ---- calculate_matrix (:124-271) -----------------------------------------------------------------------------
        const int ns = TR ? m : n, nl = TR ? n : m;  // lengths along the sweep and across the lanes
        const uint8_t *seq_s = TR ? s_alt : s_ref, *seq_l = TR ? s_ref : s_alt;
        const int my_strips = dp ? (nl + strip_cols - 1) / strip_cols : 0;
        int n_strips = my_strips, n_max = dp ? ns : 0, m_max = dp ? nl : 0;
#pragma unroll
        for (int o = 32; o >= SW_L; o >>= 1) {  // over the groups
            n_strips = max(n_strips, __shfl_xor(n_strips, o, WAVE));
            n_max = max(n_max, __shfl_xor(n_max, o, WAVE));
            m_max = max(m_max, __shfl_xor(m_max, o, WAVE));
        }
        // the last column's best cell, tracked by the lane that owns column m (`>=`: the lowest of equals, :303-309)
        // (TR: the owner of the last row; it writes the bottom row, and every lane looks at the last column when the sweep
        // reaches it)
        const int lm = ((nl - 1) % strip_cols) / K, km = (nl - 1) % K, sm = (nl - 1) / strip_cols;
        int32_t lc_score = INT32_MIN, lc_row = 0;
        uint32_t lc_g = 0u;  // (GBIT) G of that cell
        // (GBIT) where the G bit of the lane's cell in the last column (row: TR) sits: word km / 16, bit 33 - 2 (nq - km % 16)
        const int g_word = km >> 4, g_shift = 33 - 2 * (min(K - 16 * g_word, 16) - (km & 15));
        for (int s = 0; s < n_strips; ++s) {
            const bool strip_on = dp && s < my_strips;
            const int j0 = s * strip_cols + l * K;  // columns j0+1 .. j0+K
            // the row above, updated in place: a cell's diagonal term for its right neighbour is taken (one add, the same add
            // the neighbour needs anyway) before the cell overwrites its own entry, so one register set suffices
            int32_t up[K], bgv[K];
            // the lane's bases: one register each while registers allow (two waves per SIMD leave 256), four to a register
            // beyond that (compared through a byte select: 8 % slower, measured on the 8 x 19 instance)
            constexpr bool PACKED = K > 24;
            uint32_t bb4[PACKED ? (K + 3) / 4 : 1] = {}, bb1[PACKED ? 1 : K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = j0 + k + 1;
                // (columns beyond the sequence compute values nobody reads: whatever they compare with)
                const uint32_t base_k = (strip_on && j <= nl) ? (uint32_t)seq_l[j - 1] : 0u;
                if constexpr (PACKED) bb4[k / 4] |= base_k << (8 * (k % 4));
                else bb1[k] = base_k;
                up[k] = row0(j);
                bgv[k] = SW_LOW_INIT | (TG * TAG_S);
            }
            int32_t diag = row0(j0);                     // sw[i-1][j0]
            int32_t o_sw = 0, o_bgh = 0;                 // what this lane hands to its right neighbour (row of the previous step)
            uint32_t acc_c[NH] = {}, acc_e[NH] = {};     // flag words: candidate tags (shifted in from the top), gap-open bits (from the bottom)
            // GBIT: "the walk from this cell is one diagonal", a bit per column in the position of its tag's high bit (odd bit
            // positions; every shift below is even, so the even positions -- the tags' low bits, garbage here -- never mix in)
            uint32_t gw[NH];
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) gw[hh] = ~0u;   // row 0
            uint32_t g_diag = 1u << G_IN0;               // G(i-1, j0): the diagonal of the lane's first column (row 0 / column 0: 1)
            uint32_t o_g = 0u;                           // G of the lane's last column, handed to the right neighbour like o_sw
            uint32_t *bt = slab + (size_t)s * strip_stride + (size_t)lane * NW;
            // (the reference base of the NEXT step is fetched from LDS a step ahead: its latency hides behind the cells)
            int32_t a_next = (int32_t)seq_s[max(-l, 0)];
            const bool first_strip = TR ? true : s == 0;  // (the sweep along the alternate has one strip)
            // RAMP: the first SW_L - 1 steps, while lanes are still waiting for their first row -- a lane computes only
            // inside its matrix.  After that every lane computes every step, predicate-free (8 % of the kernel): rows
            // beyond the alignment's last (other alignments of the wave are longer) and strips it does not have produce
            // values nobody reads -- their flag stores land in the slab's unused part -- and only what leaves the lane's
            // registers for LDS or the best-cell bookkeeping asks `live`.
            // LEAN: every alignment of the wave has ONE strip (the usual case) -- the steady-state step then has no branch but the
            // rare last-row one: no strip edges, and the last column's cell is taken with masks instead of under a condition.
            // (The general step has 19 branches and 47 scalar instructions next to its 345 vector ones, and a lone taken
            // branch costs a wave more than the cells between two of them.)
            uint32_t kmask[K];                           // all ones for the lane's cell in the last column (row: TR), else zero
#pragma unroll
            for (int k = 0; k < K; ++k) {
                kmask[k] = k == km ? ~0u : 0u;
                asm volatile("" : "+v"(kmask[k]));       // (vector registers: as conditions they would be 2 K scalar registers, spilled)
            }
            auto g_of = [&](int k) -> uint32_t {         // (GBIT) G of the lane's column k, this row
                const int hh = k >> 4, nq = min(K - 16 * hh, 16);
                return (gw[hh] >> (33 - 2 * (nq - (k & 15)))) & 1u;
            };
            auto g_last = [&]() -> uint32_t {            // ... of its cell in the last column (a run-time position)
                uint32_t w = gw[0];
#pragma unroll
                for (int hh = 1; hh < NH; ++hh) w = g_word == hh ? gw[hh] : w;
                return (w >> g_shift) & 1u;
            };
            auto step = [&](auto ramp_c, auto lean_c, const int t) {
                constexpr bool RAMP = decltype(ramp_c)::value, LEAN = decltype(lean_c)::value;
                const int i = t - l + 1;                 // this lane's row at this step
                int32_t left = row_shr1<SW_L>(o_sw), h_bg = row_shr1<SW_L>(o_bgh);
                uint32_t g_in = 0u;
                if constexpr (GBIT) g_in = (uint32_t)row_shr1<SW_L>((int32_t)o_g) | (l == 0 ? 1u << G_IN0 : 0u);  // (column 0: 1)
                const bool live = strip_on && i >= 1 && i <= ns;
                const bool active = RAMP ? live : true;
                const int32_t a_base = a_next;
                a_next = (int32_t)seq_s[max(i, 0)];      // row i + 1 (the LDS area is padded: bytes beyond the sequence are harmless)
                if (active) {
                    if (LEAN || first_strip) {           // column 0: gap penalties (:161-168) or zeros; no horizontal gap yet
                        left = l == 0 ? (edge_gaps ? x_open + (i - 1) * x_extend : 0) : left;
                        h_bg = l == 0 ? (SW_LOW_INIT | (TG * TAG_L)) : h_bg;
                    } else if (l == 0) {                 // the right edge of the previous strip
                        left = e_sw[min(i, ns)];
                        h_bg = e_bgh[min(i, ns)];
                    }
                    const int32_t diag_next = left;      // sw[i][j0]: the diagonal of this lane's first column, next row
                    auto score = [&](int k) {
                        if constexpr (PACKED) return (uint32_t)a_base == ((bb4[k / 4] >> (8 * (k % 4))) & 0xffu) ? x_match : x_mismatch;
                        else return (uint32_t)a_base == bb1[k] ? x_match : x_mismatch;
                    };
                    int32_t step_diag = diag + score(0);                                           // :194-199 (tag: diagonal)
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int32_t pv = up[k] + x_open_s;                                       // :207-218
                        const int32_t next_diag = k + 1 < K ? up[k] + score(k + 1) : 0;           // (before up[k] becomes this row's value)
                        const int32_t ev = bgv[k] + x_extend;
                        if constexpr (!LITE) acc_e[k / 16] = __builtin_amdgcn_alignbit(acc_e[k / 16], (uint32_t)(ev - pv), 31);  // 1: pv > ev, the gap opens here
                        bgv[k] = max(pv, ev);
                        const int32_t ph = left + x_open_l;                                        // :229-240 (tag: right)
                        const int32_t eh = h_bg + x_extend;
                        if constexpr (!LITE) acc_e[k / 16] = __builtin_amdgcn_alignbit(acc_e[k / 16], (uint32_t)(eh - ph), 31);
                        h_bg = max(ph, eh);
                        // priority: diagonal, then right (horizontal), then down (:250-266) -- the tags break the ties
                        if constexpr (!WIDE) {
                            const int32_t cx = max(step_diag, max(h_bg, bgv[k]));
                            acc_c[k / 16] = __builtin_amdgcn_alignbit((uint32_t)cx, acc_c[k / 16], 2);
                            left = up[k] = cx & ~3;
                        } else {  // the reference's comparisons, then its clamp
                            const int32_t g_h = TR ? bgv[k] : h_bg, g_v = TR ? h_bg : bgv[k];  // the horizontal / vertical gap candidates
                            const int32_t gap = max(g_h, g_v);
                            const uint32_t tag = step_diag >= gap ? (uint32_t)TAG_DIAG : g_h >= g_v ? (uint32_t)TAG_RIGHT : (uint32_t)TAG_DOWN;
                            acc_c[k / 16] = __builtin_amdgcn_alignbit(tag, acc_c[k / 16], 2);
                            left = up[k] = max(max(step_diag, gap), -100000000);
                        }
                        step_diag = next_diag;
                    }
                    // (streaming stores: 0.6 bytes per cell that nobody reads before the backtrack -- the flags are a quarter of
                    // the kernel's time, in proportion to their volume)
                    // one store per lane and step: the lane's NW dwords lie next to each other ([strip][step][lane][dword]), a
                    // wave's store covers NW x 256 contiguous bytes.  (Three dword stores per step, 256 bytes apart, cost the
                    // kernel a fifth of its time: a vector-memory instruction holds up its wave's issue for ~100 clocks.)
                    if constexpr (GBIT) {
                        // G(i, k) = [tag(i, k) is DIAG] & G(i-1, k-1): the words move up one cell (two bits), the diagonal of the
                        // first column comes in at the bottom, a word's top cell goes on to the next word.  (A word of fewer than 16
                        // cells keeps older steps' tags below them: masked off before the shift.)
                        uint32_t carry = g_diag;
#pragma unroll
                        for (int hh = 0; hh < NH; ++hh) {
                            const int nq = K - 16 * hh < 16 ? K - 16 * hh : 16;
                            const uint32_t valid = nq == 16 ? ~0u : ~0u << (32 - 2 * nq);
                            const uint32_t old = gw[hh] & valid;
                            gw[hh] = ((old << 2) | carry) & acc_c[hh];
                            if (hh + 1 < NH) {
                                const int nq_next = K - 16 * (hh + 1) < 16 ? K - 16 * (hh + 1) : 16;
                                carry = old >> (31 - (33 - 2 * nq_next));   // bit 31 (the word's top cell) -> the next word's first cell
                            }
                        }
                        g_diag = g_in;                                      // G(i, j0), the first column's diagonal one row on
                        o_g = (gw[NH - 1] & 0x80000000u) >> (31 - G_IN0);   // the lane's last column, where the neighbour's first cell takes it
                    }
                    if constexpr (!LITE) {
                    uint32_t *row_bt = bt + (size_t)t * NW * WAVE;
                    typedef uint32_t flag_vec __attribute__((ext_vector_type(NW)));
                    flag_vec fv;
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh) {
                        if (LAST_PACKED && hh == NH - 1) {
                            constexpr uint32_t LO = REM >= 16 ? ~0u : (1u << (2 * (REM & 15))) - 1u;
                            fv[2 * hh] = (acc_c[hh] & ~(~0u >> (2 * (REM & 15)))) | (acc_e[hh] & LO);
                        } else {
                            fv[2 * hh] = acc_c[hh];
                            fv[2 * hh + 1] = acc_e[hh];
                        }
                    }
                    store_flags<NW>(row_bt, fv);
                    }
                    diag = diag_next;
                    o_sw = left;
                    o_bgh = h_bg;
                    if (!LEAN && !TR && live && l == SW_L - 1 && s + 1 < my_strips) {  // leaves the strip: the next one picks it up at this row
                        e_sw[i] = left;
                        e_bgh[i] = h_bg;
                    }
                    if constexpr (LEAN) {
                        // (one v_and_or_b32 per column, written out: from `v |= up[k] & kmask[k]` the compiler builds a tree of bitop3 / or3
                        // -- 30 instructions per step of 19 cells instead of 19 -- and from a select form two per column.  Measured and
                        // dropped: `v = up[km]`, a compare-and-select chain on hoisted scalar masks: 2.80 -> 2.95 ms on the 8 x 19 instance.)
                        int32_t v = 0;
                        // (four columns to a statement: the compiler keeps inline-assembly statements an s_nop apart)
                        static_for_chunks4<K>([&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            if constexpr (k + 4 <= K)
                                asm("v_and_or_b32 %0, %1, %2, %0\n\tv_and_or_b32 %0, %3, %4, %0\n\tv_and_or_b32 %0, %5, %6, %0\n\tv_and_or_b32 %0, %7, %8, %0"
                                    : "+v"(v)
                                    : "v"(up[k]), "v"(kmask[k]), "v"(up[k + 1]), "v"(kmask[k + 1]), "v"(up[k + 2]), "v"(kmask[k + 2]), "v"(up[k + 3]), "v"(kmask[k + 3]));
                            else
                                for (int q = k; q < K; ++q) v |= up[q] & (int32_t)kmask[q];
                        });
                        const bool mine = live && l == lm;
                        // (GBIT: the cell's G rides in bit 0 of what the bottom row keeps -- scores are multiples of four there)
                        if constexpr (TR) {
                            bottom[mine ? i : 0] = GBIT ? v | (int32_t)g_last() : v;    // the last row, column by column (entry 0 is nobody's)
                        } else {
                            const bool take = mine && v >= lc_score;
                            lc_score = take ? v : lc_score;
                            lc_row = take ? i : lc_row;
                            if constexpr (GBIT) lc_g = take ? g_last() : lc_g;
                        }
                    } else {
                        if (live && s == sm && l == lm) {
                            int32_t v = 0;
#pragma unroll
                            for (int k = 0; k < K; ++k) v |= up[k] & (int32_t)kmask[k];
                            if constexpr (TR) {
                                bottom[i] = GBIT ? v | (int32_t)g_last() : v;               // the last row, column by column
                            } else if (v >= lc_score) {
                                lc_score = v;
                                lc_row = i;
                                if constexpr (GBIT) lc_g = g_last();
                            }
                        }
                    }
                    if (live && i == ns) {
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            if constexpr (TR) {          // the last column: this lane's rows, top down (`>=`, as above)
                                if (j0 + k + 1 <= nl && up[k] >= lc_score) {
                                    lc_score = up[k];
                                    lc_row = j0 + k + 1;
                                    if constexpr (GBIT) lc_g = g_of(k);
                                }
                            } else if (j0 + k + 1 <= nl) {
                                bottom[j0 + k + 1] = GBIT ? up[k] | (int32_t)g_of(k) : up[k];
                            }
                        }
                    }
                }
            };
            // the sweep ends when the last lane that owns columns has done its last row (a 150-base read on 64 lanes of
            // three columns: 50 lanes); rounded up to even (the extra step is nobody's row)
            const int lanes_in_use = min(SW_L, (m_max - s * strip_cols + K - 1) / K);
            const int steps = (n_max + lanes_in_use) & ~1;
            constexpr int RAMP_STEPS = SW_L & ~1;       // (even: the loops take two steps at a time)
            for (int t = 0; t < min(RAMP_STEPS, steps); t += 2) {
                step(std::true_type{}, std::false_type{}, t);
                step(std::true_type{}, std::false_type{}, t + 1);
            }
            if (n_strips == 1) {
                for (int t = RAMP_STEPS; t < steps; t += 2) {
                    step(std::false_type{}, std::true_type{}, t);
                    step(std::false_type{}, std::true_type{}, t + 1);
                }
            } else {
                for (int t = RAMP_STEPS; t < steps; t += 2) {
                    step(std::false_type{}, std::false_type{}, t);
                    step(std::false_type{}, std::false_type{}, t + 1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // ---- where the backtrack starts (:289-330) ---------------------------------------------------------------------
        Start best{INT32_MIN, 0, 0, 0, 0};
        int32_t segment_length = 0;
        if (dp) {
            if (p.strategy == PHMM_SW_STRATEGY_INDEL) {
                best = Start{0, 0, 0, n, m};
            } else {
                // the owner of the last column holds its best cell; everybody gets it
                const int src = (lane & GMASK) | lm;
                int32_t sc = __shfl(lc_score, src, WAVE), rw = __shfl(lc_row, src, WAVE);
                uint32_t gg = GBIT ? (uint32_t)__shfl((int)lc_g, src, WAVE) : 0u;
                if constexpr (TR) {  // every lane holds the best of its rows: the highest, among equals the lowest row down
                    sc = lc_score;
                    rw = lc_row;
                    gg = lc_g;
#pragma unroll
                    for (int o = SW_L / 2; o >= 1; o >>= 1) {
                        const int32_t s2 = __shfl_xor(sc, o, WAVE), r2 = __shfl_xor(rw, o, WAVE);
                        const uint32_t g2 = GBIT ? (uint32_t)__shfl_xor((int)gg, o, WAVE) : 0u;
                        if (s2 > sc || (s2 == sc && r2 > rw)) {
                            sc = s2;
                            rw = r2;
                            gg = g2;
                        }
                    }
                }
                best = Start{sc, abs(rw - m), 0, rw, m, gg};
                if (p.strategy != PHMM_SW_STRATEGY_LEADING_INDEL) {
                    for (int j = l + 1; j <= m; j += SW_L) {  // bottom row, every lane a share of the columns
                        const int32_t bj = bottom[j];
                        const Start c{GBIT ? bj & ~3 : bj, abs(n - j), j, n, j, GBIT ? (uint32_t)bj & 1u : 0u};
                        if (better(c, best)) best = c;
                    }
                }
            }
        }
        if (p.strategy != PHMM_SW_STRATEGY_INDEL && p.strategy != PHMM_SW_STRATEGY_LEADING_INDEL) {
#pragma unroll
            for (int o = SW_L / 2; o >= 1; o >>= 1) {  // best of the group
                Start c;
                c.score = __shfl_xor(best.score, o, WAVE);
                c.dist = __shfl_xor(best.dist, o, WAVE);
                c.order = __shfl_xor(best.order, o, WAVE);
                c.p1 = __shfl_xor(best.p1, o, WAVE);
                c.p2 = __shfl_xor(best.p2, o, WAVE);
                if constexpr (GBIT) c.g = (uint32_t)__shfl_xor((int)best.g, o, WAVE);
                if (better(c, best)) best = c;
            }
        }
        if (dp && best.order > 0) segment_length = m - best.p2;  // a bottom-row cell: the end of the alternate overhangs (:327)
        // Every lane's backtrack entries have to be visible to the lanes that walk them: the wave's own stores are complete
        // (acknowledged by L2) once its vector-memory counter is zero, and the walk reads them with device-scope loads (L2, not
        // this CU's L1, which may still hold the slab's lines of the previous round).  (A device-scope fence here -- write back
        // and invalidate -- cost 5 % of the kernel.)
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)

        // ---- calculate_cigar (:332-443): the sixteen lanes of the alignment walk together ------------------------------
        // Every backtrack step is a dependent read from HBM; sixteen cells down the diagonal are fetched at once, one
        // per lane, and the run of diagonal steps among them is taken in one go -- a read of 150 bases is traced in a
        // dozen round trips instead of 150.  Gap cells (rare) are handled one at a time, every lane doing the same.
        if (valid) {
            CigarOut cig{p.cigar + (p.cigar_off ? p.cigar_off[a] : (uint64_t)a * p.cigar_slot),
                         p.cigar_off ? p.cigar_off[a + 1] - p.cigar_off[a] : (uint64_t)p.cigar_slot, l == 0};
            int32_t alignment_offset = 0;
            bool again = false;  // (tags-only sweep) the walk met a gap
            if (n == 0 || m == 0) {  // the reference asserts (:65-68, :132-134); the host refuses such input beforehand
                if (l == 0) p.status[SW_STATUS_EMPTY] = 1u;
            } else if (found >= 0) {
                cig.push(make_element(ST_MATCH, (uint32_t)m));
                alignment_offset = found;
            } else {
                // flag words of cell (i, jj): [0] candidate tags, [eo] gap-open bits (eo = 0 where the two share a dword);
                // `sh` = 2 x (cells after it in the word)
                auto cell_words = [&](int i, int jj, int &sh, int &eo) -> const uint32_t * {
                    if constexpr (TR) {  // rows across the lanes, columns along the sweep
                        const int t2 = i;
                        i = jj;
                        jj = t2;
                    }
                    const int ss = (jj - 1) / strip_cols, cc = (jj - 1) % strip_cols, ll = cc / K, kk = cc % K;
                    const int hh = kk >> 4, nq = min(K - 16 * hh, 16);
                    sh = 2 * (nq - 1 - (kk & 15));
                    eo = LAST_PACKED && hh == NH - 1 ? 0 : 1;
                    return slab + (size_t)ss * strip_stride + ((size_t)(i - 1 + ll) * WAVE + (lane & GMASK) + ll) * NW + (LITE ? hh : 2 * hh);
                };
                constexpr int BQ = SW_L >= 32 ? 1 : 32 / SW_L;  // cells a lane fetches per round trip of the walk
                int p1 = best.p1, p2 = best.p2;
                if (segment_length > 0 && p.strategy == PHMM_SW_STRATEGY_SOFTCLIP) {
                    cig.push(make_element(ST_CLIP, (uint32_t)segment_length));
                    segment_length = 0;
                }
                int state = ST_MATCH;
                constexpr int HB = TR ? 1 : 0, VB = TR ? 0 : 1;  // the sweep's gap is shifted in first, the lanes' second
                // The start cell's G bit says whether the reference's walk from it (:372-417) takes the diagonal all the way to row 0 /
                // column 0 -- then it is `run` times the loop body with btrack == 0, and no flag is read -- or meets a gap somewhere:
                // the tags-only sweep (which stored no flags) then hands the alignment to the full instance, the full instance walks.
                bool diagonal = false;
                if constexpr (GBIT)
                    diagonal = best.g && my_strips == 1 && (p.strategy == PHMM_SW_STRATEGY_SOFTCLIP || p.strategy == PHMM_SW_STRATEGY_IGNORE);
                if (diagonal) {
                    const int run = min(p1, p2);
                    segment_length += run;
                    p1 -= run;
                    p2 -= run;
                } else if constexpr (LITE) {
                    again = true;
                }
                for (; !LITE && !diagonal;) {
                    // lane l looks at the cells (p1 - d, p2 - d), d = l, l + SW_L, ... (BQ of them, 32 cells per group and round trip:
                    // every fetch is a dependent read from HBM and the run of diagonal steps is usually the whole read);
                    // `run` = diagonal steps from (p1, p2) before anything else
                    int sh, eo;
                    uint32_t tags[BQ];
#pragma unroll
                    for (int q = 0; q < BQ; ++q) {
                        const int d = l + q * SW_L;
                        const bool inside = p1 - d >= 1 && p2 - d >= 1;
                        const uint32_t *w = cell_words(inside ? p1 - d : 1, inside ? p2 - d : 1, sh, eo);
                        const uint32_t word = flag_load(w);
                        tags[q] = inside ? (word >> (30 - sh)) & 3u : 3u;
                    }
                    int run = BQ * SW_L;
                    uint32_t tag = 3u;  // of the cell the run stops at (fetched by lane run % SW_L as its cell run / SW_L)
#pragma unroll
                    for (int q = BQ - 1; q >= 0; --q) {
                        const uint64_t others = ~(__ballot(tags[q] == TAG_DIAG) >> (lane & GMASK)) & LMASK;  // lanes that do not see a diagonal step
                        if (others) {
                            run = q * SW_L + __ffsll((long long)others) - 1;
                            tag = tags[q];
                        }
                    }
                    if (run > 0) {  // `run` times the reference's loop body with btrack == 0 (:372-417)
                        if (state != ST_MATCH) {
                            if (segment_length > 0) cig.push(make_element(state, (uint32_t)segment_length));
                            segment_length = 0;
                            state = ST_MATCH;
                        }
                        segment_length += run;
                        p1 -= run;
                        p2 -= run;
                        if (p1 <= 0 || p2 <= 0) break;
                        if (run == BQ * SW_L) continue;
                    }
                    // a gap ends at (p1, p2).  The reference's btrack entry (:257-266) is +k (k rows up) or -k (k columns
                    // left), k = the length the best gap ending here has: 1 where it opens, else one more than at the
                    // previous cell of the column / row
                    const int src = (lane & GMASK) | (run & (SW_L - 1));  // the lane that fetched this cell
                    const uint32_t gtag = (uint32_t)__shfl((int)tag, src, WAVE);
                    const uint32_t *w = cell_words(p1, p2, sh, eo);
                    uint32_t e = flag_load(w + eo);
                    int32_t k = 1;
                    if (gtag == TAG_RIGHT) {
                        for (int j2 = p2; !((e >> (sh + HB)) & 1u) && j2 > 1;) {
                            ++k;
                            --j2;
                            e = flag_load(cell_words(p1, j2, sh, eo) + eo);
                        }
                        p2 -= k;
                    } else {
                        for (int i2 = p1; !((e >> (sh + VB)) & 1u) && i2 > 1;) {
                            ++k;
                            --i2;
                            e = flag_load(cell_words(i2, p2, sh, eo) + eo);
                        }
                        p1 -= k;
                    }
                    const int new_state = gtag == TAG_RIGHT ? ST_INSERTION : ST_DELETION;
                    if (new_state == state) {
                        segment_length += k;
                    } else {
                        if (segment_length > 0) cig.push(make_element(state, (uint32_t)segment_length));
                        segment_length = k;
                        state = new_state;
                    }
                    if (p1 <= 0 || p2 <= 0) break;
                }
                if (again) {
                } else if (p.strategy == PHMM_SW_STRATEGY_SOFTCLIP) {
                    cig.push(make_element(state, (uint32_t)segment_length));
                    if (p2 > 0) cig.push(make_element(ST_CLIP, (uint32_t)p2));
                    alignment_offset = p1;
                } else if (p.strategy == PHMM_SW_STRATEGY_IGNORE) {
                    cig.push(make_element(state, (uint32_t)(segment_length + p2)));
                    alignment_offset = p1 - p2;
                } else {
                    cig.push(make_element(state, (uint32_t)segment_length));
                    if (p1 > 0)
                        cig.push(make_element(ST_DELETION, (uint32_t)p1));
                    else if (p2 > 0)
                        cig.push(make_element(ST_INSERTION, (uint32_t)p2));
                    alignment_offset = 0;
                }
            }
            if (l == 0 && again) {
                p.todo_out[atomicAdd(p.todo_out_count, 1u)] = a;
                p.n_cigar[a] = 0;
                p.alignment_offset[a] = 0;
            } else if (l == 0) {
                cig.finish();
                p.n_cigar[a] = cig.n;
                p.alignment_offset[a] = alignment_offset;
                if (cig.n > cig.cap) p.status[SW_STATUS_CAPACITY] = 1u;
            }
        }
        __builtin_amdgcn_s_barrier();  // (one wave per block: a scheduling point between rounds)
    }
    // (before the block counts itself in: once the count is complete the caller may have its results and the status block's
    // memory -- the pinned mirror, for a call that aligns every pair -- may hold the next call's inputs.  These two words,
    // stored behind the count, landed in a later call's read bases once in a few hundred calls: tools/threads_bench TB_VERIFY.
    // Since round 5 they are measurement only: no production launch stores them, and phmm_region.cpp never asks.)
    if (p.report_clock == 1u && vblock == 0 && lane == 0) {
        p.status[2] = (uint32_t)(clock64() - clk0);        // shader clocks
        p.status[3] = (uint32_t)(wall_clock64() - wall0);  // 100 MHz ticks
    }
    if (p.done_counter) {  // (one wave per block: its stores are behind the fence, then it is counted)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_fetch_add(p.done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (tests only, report_clock == 2: round 4's bug on purpose -- two words stored BEHIND the count, ~300 us late, i.e. when the
    // caller may already have staged its next call where this call's status block was.  tests/test_region_handoffs.py shows
    // that PHMM_MIRROR_CANARY turns such a store into a failed call.)
    // (EVERY block: the one that completes the count is then ~300 us late for sure)
    if (p.report_clock == 2u && lane == 0) {
        const long long t_late = wall_clock64();
        while (wall_clock64() - t_late < 30000) __builtin_amdgcn_s_sleep(32);
        p.status[2] = 0xdeadbeefu;
        p.status[3] = 0xdeadbeefu;
    }
}

}  // namespace swdev

}  // namespace phmm
