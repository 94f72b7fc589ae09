// gfx950 chained forward kernel, single precision first -- the OPT-IN mode PHMM_FLAG_F32_FIRST.
//
// The reference's production arm (gkl, Cargo.toml:42; called at src/pair_hmm/pair_hmm.rs:345-375) computes every
// pair in f32 with a 2^120 scale first and recomputes it in f64 only when the f32 result is too small to trust.
// This file is that scheme for the chained sweep of phmm_chain_kernels.hip (16 or 32 lanes per pair, one compilation
// unit each: -DPHMM_CHAIN32_L=16|32):
//   * same stream-of-rows design, same folded 7-instruction cell, state and row constants in f32 (the constants are
//     still derived in f64 from the f64 tables and rounded once when a row record is written);
//   * half the state registers and a 36-byte row record: three waves per SIMD instead of two (152 VGPRs at K = 19),
//     and the f32 form of the cell issues at ~3 clk per instruction instead of ~4.7 (tools/ubench/issue.hip);
//   * every haplotype starts from D(0,j) = 2^100; a pair whose scaled row sum comes out below 2^-96 (or not finite,
//     or whose log10 would be positive), i.e. likelihood x haplotype length < 2^-196 ~ 1e-59, is not trusted (every
//     term that matters for a larger sum is >= 2^-120, still a normal f32; gkl draws its line at a scaled 1e-28 under
//     2^120, i.e. ~1e-64): the kernel sets redo[read] and the host
//     enqueues the f64 per-read kernel (phmm_forward<16,K>) right behind it, which recomputes exactly the flagged
//     reads -- all their haplotypes -- and overwrites their results;
//   * runs that need the general path (a haplotype with 'N', a gcp == 0, a base quality 0) are flagged wholesale
//     and left to that f64 kernel, so there is no second code path in here.
// Results of unflagged pairs differ from the f64 path by f32 rounding (measured <= 2e-6 in log10 on the reference's
// known-answer vectors and on synthetic regions; the reference's own gate is 1e-5); flagged pairs are the f64 results.
#include "phmm_device.hpp"

namespace phmm {

namespace {

#ifndef PHMM_CHAIN32_L
#define PHMM_CHAIN32_L 16
#endif
constexpr int CL = PHMM_CHAIN32_L;    // lanes per pair
constexpr int RING = 256;             // ring rows (power of two), shared by the streams
constexpr int RING_SLOTS = RING + 4 + 1;  // every stream's rows + a guard slot repeating its row 0 (phmm_chain_kernels.hip), then the neutral row
constexpr int CHAIN_META = CHAIN_MAX_READS + 8;
constexpr uint32_t X_PAD = 0x100u;    // base code of padding columns (>= H) and read-side code of the SUM row
constexpr uint32_t X_NONE = 0x102u;   // read-side code that matches nothing
constexpr int LEAD = CL - 1;

struct alignas(4) Row32 {  // 36 bytes: an odd dword stride, conflict-free for the staggered per-lane reads
    float mm, bI, gI, dDp, dd, px;
    uint32_t x, pad0;  // read base; SUM rows: read index inside its stream
    float inj;         // RESET rows: D'(0,.) of the next read, injected at the group's first lane
};
static_assert(sizeof(Row32) == 36, "LDS row record (f32)");

__device__ __forceinline__ Row32 lds_row32(const Row32 *rows, int idx) {
    return *reinterpret_cast<const Row32 *>(reinterpret_cast<const unsigned char *>(rows) + __mul24(idx, (int)sizeof(Row32)));
}

__device__ __forceinline__ Row32 lds_row32_at(uint32_t at, int skip) {  // the record at LDS byte address `at` (+ `skip` records)
    Row32 r;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(3))) uint32_t *LdsU32;
    const LdsU32 w = (LdsU32)(uintptr_t)at + skip * (int)(sizeof(Row32) / 4);
    uint32_t v[sizeof(Row32) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Row32) / 4); ++i) v[i] = w[i];
    __builtin_memcpy(&r, v, sizeof r);
#else
    (void)at;
    (void)skip;
    r = Row32{};
#endif
    return r;
}

// lane n <- lane n-1 inside each group; the group's first lane gets `inject` (0 for M and I): through the DPP `old`
// operand where the shift has no source lane (row_shr:1 for 16-lane groups, wave_shr:1 for lane 0), through a select
// for lane 32 of two 32-lane groups
__device__ __forceinline__ float from_left_inject32(float v, float inject, bool group_head) {
    constexpr int ctrl = CL == 16 ? 0x111 : 0x138;
    float r = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(inject), __float_as_int(v), ctrl, 0xf, 0xf, false));
    if constexpr (CL == 32) r = group_head ? inject : r;
    return r;
}
__device__ __forceinline__ float from_left32(float v, bool group_head) { return from_left_inject32(v, 0.f, group_head); }

// One read row for the K columns of a lane, f32, every register updated in place.  Same cell, same fixed order as
// row_update<K, ROW_FAST_EXEC> (phmm_device.hpp) -- the compiler's own schedule of this loop is 1.75x slower -- except
// for the prior select: a 32-bit value needs only ONE v_cndmask, and compare + select + multiply (three plain
// instructions, ~2 clk each) beat the EXEC-masked multiply, whose EXEC round trip costs the f32 cell more than it saves
// (tools/ubench/issue.hip: 17 vs 23 clk per cell at two waves; +5 % in the kernel) -- the opposite of the f64 case,
// where the select needs two v_cndmask and the arithmetic is slower anyway.
//   M~(k)  = D'(k-1)*dDp + I^(k-1);  M~(k) += M~(k-1)*mm;  M~(k) *= (x != y_k ? px : 1)
//   I^(k-1) = I^(k-1)*gI + M~(k-1)*bI;                       then the serial chain D'(k) = D'(k-1)*dd + M~(k-1)
// Haplotype columns: two 16-bit codes per dword for K > 21 (compared through an SDWA half-word select: keeps K = 22..25
// at three waves per SIMD), one per dword otherwise.
template <int K>
__device__ __forceinline__ void row_update32(float (&Mp)[K], float (&Ip)[K], float (&Dp)[K], const float plM, const float plI,
                                             const float plD, const float lM, const float lD, const Row32 &c,
                                             const HapCols<K> &hc) {
    Ip[K - 1] = fmaf(Mp[K - 1], c.bI, Ip[K - 1] * c.gI);
    static_for_down<K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr bool packed = K > PHMM_SDWA_MIN_K;
        const uint32_t yk = packed ? hc.y[k >> 1] : (uint32_t)hc.base(k);
        constexpr int km1 = k > 0 ? k - 1 : 0;
        float sel;  // 1 where the bases match, px (= mismatch / match prior) where they do not
#define PHMM_CELL32(CMP)                                                                                            \
    asm volatile("v_fma_f32 %[M], %[Dl], %[dDp], %[Il]\n\t"                                                        \
                 "v_fma_f32 %[M], %[Ml], %[mm], %[M]\n\t"                                                          \
                 CMP "\n\t"                                                                                      \
                 "v_cndmask_b32_e32 %[sel], 1.0, %[px], vcc\n\t"                                                   \
                 "v_mul_f32 %[M], %[sel], %[M]\n\t"                                                                \
                 "v_mul_f32 %[Il], %[Il], %[gI]\n\t"                                                               \
                 "v_fma_f32 %[Il], %[Ml], %[bI], %[Il]"                                                             \
                 : [M] "=&v"(Mp[k]), [Il] "+v"(Ip[km1]), [sel] "=&v"(sel)                                           \
                 : [Dl] "v"(Dp[km1]), [dDp] "v"(c.dDp), [Ml] "v"(Mp[km1]), [mm] "v"(c.mm), [x] "v"(c.x), [y] "v"(yk), \
                   [px] "v"(c.px), [gI] "v"(c.gI), [bI] "v"(c.bI)                                                    \
                 : "vcc")
        if constexpr (k > 0 && packed && (k & 1)) {
            PHMM_CELL32("v_cmp_ne_u32_sdwa vcc, %[x], %[y] src0_sel:DWORD src1_sel:WORD_1");
        } else if constexpr (k > 0 && packed) {
            PHMM_CELL32("v_cmp_ne_u32_sdwa vcc, %[x], %[y] src0_sel:DWORD src1_sel:WORD_0");
        } else if constexpr (k > 0) {
            PHMM_CELL32("v_cmp_ne_u32_e32 vcc, %[x], %[y]");
#undef PHMM_CELL32
        } else {
            const float a0 = fmaf(plM, c.mm, fmaf(plD, c.dDp, plI));
            Mp[0] = (c.x != (uint32_t)hc.base(0)) ? a0 * c.px : a0;
        }
    });
    float leftM = lM, leftD = lD;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Dp[k] = fmaf(leftD, c.dd, leftM);
        leftM = Mp[k];
        leftD = Dp[k];
    }
}

__device__ __forceinline__ Row32 neutral_row32() {
    Row32 n;
    n.mm = 0.f; n.bI = 0.f; n.gI = 1.f; n.dDp = 0.f; n.dd = 1.f; n.px = 0.f;
    n.x = 0; n.pad0 = 0; n.inj = 0.f;
    return n;
}

}  // namespace

// One work item = one wave (see phmm_chain_kernels.hip: every item carries its K and stream count).
template <int CLT, int K>  // CLT == CL of this compilation unit (keeps the units' kernel symbols apart)
__device__ __forceinline__ void chain_body_f32(const ChainParams &cp, const ChainItem it, unsigned char *smem) {
    static_assert(CLT == CL, "one lanes-per-pair value per compilation unit");
    const ForwardParams &p = cp.f;
    const int lane = threadIdx.x;
    const int grp = lane / CL, l = lane % CL;
    const bool group_head = (CL == 32) && (lane == 32);
    const uint32_t reg = it.region;
    const int n_chain = (int)(it.read_end - it.read_begin);
    const uint32_t h0 = p.region_hap_off[reg];
    const int Nh = (int)(p.region_hap_off[reg + 1] - h0);
    const int S = (CL == 16) ? (int)it.streams : 1;
    const int GS = (WAVE / CL) / S;
    const int sid = grp / GS;
    const int a = (int)it.quad * GS + grp % GS;
    const bool hv = a < Nh;
    const int n_sub = (n_chain + S - 1) / S;
    const int TPS = WAVE / S;
    const int NM = RING / S - 1;
    auto n_of = [&](int s) { return max(0, min(n_sub, n_chain - s * n_sub)); };
    const int n_mine = n_of(sid);
    uint32_t ho = 0;
    int H = 0;
    if (hv) {
        ho = p.hap_off[h0 + a];
        H = (int)(p.hap_off[h0 + a + 1] - ho);
    }
    Row32 *ring = reinterpret_cast<Row32 *>(smem);                     // RING_SLOTS records
    uint32_t *roff = reinterpret_cast<uint32_t *>(ring + RING_SLOTS);  // per stream s at s*(n_sub+1): byte offset of each read
    uint32_t *stot = roff + CHAIN_META;                                // [4] rows of each stream

    HapCols<K> hc;
    bool lane_n = false;
#pragma unroll
    for (int w = 0; w < HapCols<K>::W; ++w) hc.y[w] = 0u;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int col = l * K + k;
        const uint32_t yv = col < H ? (uint32_t)p.hap_bases[ho + col] : X_PAD;
        lane_n |= (yv == 'N');
        hc.set(k, yv);
    }

    const uint32_t rb = it.read_begin;
    const uint32_t byte0 = p.read_off[rb];
    const uint32_t bytes = p.read_off[it.read_end] - byte0;
    bool z = false;
    for (uint32_t i = lane; i < bytes; i += WAVE) z |= row_blocks_prescale(p, byte0 + i);
    if ((__ballot(z) | __ballot(lane_n)) != 0ull) {  // general path needed: leave the whole run to the f64 kernel
        if (lane < n_chain) cp.redo[rb + lane] = 1;
        return;
    }
    {
        const int sj = lane / n_sub, ij = lane % n_sub;
        uint32_t len = lane < n_chain ? p.read_off[rb + lane + 1] - p.read_off[rb + lane] + 2u : 0u;  // + SUM + RESET
        uint32_t incl = len;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += v;
        }
        const uint32_t before = __shfl(incl, max(sj * n_sub - 1, 0), WAVE);
        const int cb = sj * (n_sub + 1);
        if (lane < 4) stot[lane] = 0u;
        if (lane < n_chain) {
            roff[cb + ij] = p.read_off[rb + lane];
            if (ij + 1 == n_of(sj)) {
                roff[cb + ij + 1] = p.read_off[rb + lane + 1];
                stot[sj] = incl - (sj > 0 ? before : 0u);
            }
        }
        if (lane == 0) ring[RING_SLOTS - 1] = neutral_row32();
    }
    lds_wave_sync();
    const int S_max = (int)max(max(stot[0], stot[1]), max(stot[2], stot[3]));
    const float c_unit = 0x1p100f;  // common D(0,j) of every haplotype

    // ---- row producer (see phmm_chain_kernels.hip): constants in f64 from the f64 tables, rounded once -------
    uint32_t pb_x = 0, pb_q = 0, pb_qp = 0, pb_i = 0, pb_d = 0, pb_dp = 0, pb_g = 0, pb_gn = 0;
    const int ps = lane / TPS, pj = lane % TPS;
    const int pcb = ps * (n_sub + 1), pn = n_of(ps);
    int p_lo = 0, p_row = pj - LEAD - TPS;
    uint32_t p_ro = pn > 0 ? roff[pcb] : 0u;
    int p_R = pn > 0 ? (int)(roff[pcb + 1] - p_ro) : 0;
    auto advance = [&]() {
        p_row += TPS;
        while (p_lo < pn && p_row >= p_R + 2) {
            p_row -= p_R + 2;
            ++p_lo;
            if (p_lo < pn) {
                p_ro = roff[pcb + p_lo];
                p_R = (int)(roff[pcb + p_lo + 1] - p_ro);
            }
        }
    };
    auto issue = [&]() {
        advance();
        const int row = p_row, R = p_R;
        const uint32_t ro = p_ro;
        if (p_lo < pn && row >= 0 && row <= R) {
            pb_qp = row > 0 ? (uint32_t)p.base_q[ro + row - 1] : 0u;
            if (row < R) {
                pb_x = p.read_bases[ro + row];
                pb_q = p.base_q[ro + row];
                pb_i = p.ins_q[ro + row];
                pb_d = p.del_q[ro + row];
                pb_dp = row > 0 ? (uint32_t)p.del_q[ro + row - 1] : 0u;
                pb_g = p.gcp[ro + row];
                pb_gn = row + 1 < R ? (uint32_t)p.gcp[ro + row + 1] : 0u;
            }
        }
    };
    auto finish = [&](int Q0) {
        const int Q = Q0 + pj;
        const int lo = p_lo, row = p_row, R = p_R;
        Row32 n;
        if (lo < pn && row >= 0) {
            if (row < R) {
                const RowConst d = make_row_bytes(p, pb_x, pb_q, pb_qp, pb_i, pb_d, pb_dp, pb_g, pb_gn, row == 0, row + 1 >= R, true);
                n.mm = (float)d.mm; n.bI = (float)d.bI; n.gI = (float)d.gI; n.dDp = (float)d.dDp; n.dd = (float)d.dd;
                n.px = (float)d.px; n.x = d.x; n.pad0 = 0; n.inj = 0.f;
            } else if (row == R) {  // SUM row
                n.mm = (float)(1.0 - p.eps[pb_qp]); n.bI = n.mm; n.gI = 1.f; n.dDp = 0.f; n.dd = 1.f; n.px = 1.f;
                n.x = X_PAD; n.pad0 = (uint32_t)lo; n.inj = 0.f;
            } else {                // RESET row
                n.mm = 0.f; n.bI = 0.f; n.gI = 0.f; n.dDp = 0.f; n.dd = 1.f; n.px = 0.f;
                n.x = X_NONE; n.pad0 = 0;
                n.inj = c_unit * (float)(lo + 1 < pn ? 1.0 - p.eps[p.gcp[roff[pcb + lo + 1]]] : 1.0);
            }
        } else {
            n = neutral_row32();
        }
        const int slot = ps * (NM + 2) + (Q & NM);
        ring[slot] = n;
        if ((Q & NM) == 0) ring[slot + NM + 1] = n;  // the stream's guard slot
    };
    issue();
    finish(0);
    issue();
    finish(TPS);
    issue();
    finish(2 * TPS);
    lds_wave_sync();
    issue();

    // ---- state ---------------------------------------------------------------------------------------
    const float c0 = c_unit * (float)(n_mine > 0 ? 1.0 - p.eps[p.gcp[roff[sid * (n_sub + 1)]]] : 1.0);
    float Mp[K], Ip[K], Dp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Mp[k] = 0.f;
        Ip[k] = 0.f;
        Dp[k] = c0;
    }
    float aM, aI, aD, bM = 0.f, bI = 0.f, bD = c0;
    const int edge_lane = (H > 0 ? H - 1 : 0) / K, edge_k = (H > 0 ? H - 1 : 0) % K;
    const bool last_lane = (l == edge_lane);
    const uint32_t sum_code = last_lane ? X_PAD : 0xffffffffu;
    const double log10_scale = 100.0 * 0.30102999566398119521 + log10((double)H);  // log10(2^100 * H)
    auto emit = [&](const Row32 &c) {
        if (last_lane && c.x == X_PAD && hv) {
            const uint32_t r = rb + (uint32_t)(sid * n_sub) + c.pad0;
            int ek = edge_k;
            asm volatile("" : "+v"(ek));
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k == ek) sum = (Dp[k] + Mp[k]) + Ip[k];
            // too small (or not a number) to trust in f32: the f64 kernel behind this one redoes the read
            const double v = log10((double)sum) - log10_scale;
            if (!(sum >= 0x1p-96f && sum < __builtin_huge_valf()) || !(v <= 0.0)) cp.redo[r] = 1;
            p.out[p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh + a] = v;
        }
    };

    int q = LEAD - l;
    const int my_ring = sid * (NM + 2);
    const uint32_t my_rows = (uint32_t)(uintptr_t)(ring + my_ring);  // (the low half of a flat address into LDS is the LDS address)
    Row32 cA = lds_row32(ring, my_ring + (q & NM)), cB;
    int q1 = q + 1;
    const int T = (S_max + CL - 1 + 1) & ~1;
    const int phase = (int)((blockIdx.x * 2654435761u) >> 26) & (TPS - 2);
    for (int t0 = 0, t1 = min(T, TPS - phase), tick = 0; t0 < T; t0 = t1, t1 = min(T, t1 + TPS), ++tick) {
        if (tick >= 1) {
            finish(TPS * tick + 2 * TPS);
            lds_wave_sync();
            issue();
        }
        for (int t = t0; t < t1; t += 2) {
            // rows q + 1 and q + 2 from one address (the guard slot repeats row 0 behind the stream's last row)
            uint32_t pair_at;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(pair_at) : "v"((uint32_t)(q1 & NM)), "s"((uint32_t)sizeof(Row32)), "v"(my_rows));
            cB = lds_row32_at(pair_at, 0);
            aM = from_left32(Mp[K - 1], group_head);
            aI = from_left32(Ip[K - 1], group_head);
            aD = from_left_inject32(Dp[K - 1], cA.inj, group_head);
            row_update32<K>(Mp, Ip, Dp, bM, bI, bD, aM, aD, cA, hc);
            if (__ballot(cA.x == sum_code) != 0ull) emit(cA);
            cA = lds_row32_at(pair_at, 1);
            bM = from_left32(Mp[K - 1], group_head);
            bI = from_left32(Ip[K - 1], group_head);
            bD = from_left_inject32(Dp[K - 1], cB.inj, group_head);
            row_update32<K>(Mp, Ip, Dp, aM, aI, aD, bM, bD, cB, hc);
            if (__ballot(cB.x == sum_code) != 0ull) emit(cB);
            q1 += 2;
        }
    }
}

// ---- kernels: one launch per lanes-per-pair value for a mixed batch, the body alone for a uniform one ---------------
#define PHMM_CHAIN32_K_LIST(X) \
    X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) \
    X(23) X(24) X(25)

template <int CLT>
__global__ __launch_bounds__(WAVE, 2) void phmm_forward_chain_f32_any(const ChainParams cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ChainItem it = cp.items[blockIdx.x];
    switch (__builtin_amdgcn_readfirstlane((int)it.k)) {
#define PHMM_CASE(KK)                            \
    case KK:                                     \
        chain_body_f32<CLT, KK>(cp, it, smem);   \
        break;
        PHMM_CHAIN32_K_LIST(PHMM_CASE)
#undef PHMM_CASE
        default:
            break;
    }
}

template <int CLT, int K>
__global__ __launch_bounds__(WAVE, 2) void phmm_forward_chain_f32(const ChainParams cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    chain_body_f32<CLT, K>(cp, cp.items[blockIdx.x], smem);
}

#define PHMM_C32_CAT2(a, b) a##b
#define PHMM_C32_CAT(a, b) PHMM_C32_CAT2(a, b)
#define PHMM_C32_LAUNCH PHMM_C32_CAT(launch_chain_f32_L, PHMM_CHAIN32_L)

// single_k: the K every item of the launch has, or 0 for a mixed launch
hipError_t PHMM_C32_LAUNCH(int single_k, const ChainParams &cp, hipStream_t stream) {
    const size_t lds = (size_t)RING_SLOTS * sizeof(Row32) + (CHAIN_META + 4) * sizeof(uint32_t);
#define PHMM_CASE(KK)                                                                                         \
    if (single_k == KK) {                                                                                     \
        hipLaunchKernelGGL((phmm_forward_chain_f32<CL, KK>), dim3(cp.n_items), dim3(WAVE), lds, stream, cp);   \
        return hipGetLastError();                                                                             \
    }
    PHMM_CHAIN32_K_LIST(PHMM_CASE)
#undef PHMM_CASE
    hipLaunchKernelGGL((phmm_forward_chain_f32_any<CL>), dim3(cp.n_items), dim3(WAVE), lds, stream, cp);
    return hipGetLastError();
}

#if PHMM_CHAIN32_L == 16
hipError_t launch_chain_f32_L32(int single_k, const ChainParams &cp, hipStream_t stream);
hipError_t launch_chain_f32(int L, int single_k, const ChainParams &cp, hipStream_t stream) {
    if (!cp.n_items) return hipSuccess;
    return L == 16 ? launch_chain_f32_L16(single_k, cp, stream) : L == 32 ? launch_chain_f32_L32(single_k, cp, stream)
                                                                           : hipErrorInvalidValue;
}
#endif

}  // namespace phmm
