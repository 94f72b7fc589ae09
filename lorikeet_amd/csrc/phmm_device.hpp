// Device-side building blocks shared by the forward kernels (phmm_kernels.hip, phmm_chain_kernels.hip):
// DPP lane shifts, the LDS row record, packed haplotype columns, the in-place row update and the sweeps.
// See the header comment of phmm_kernels.hip for the mapping.
#pragma once
#include "phmm_internal.hpp"

#include <type_traits>

// K up to this value is compiled for two resident waves per SIMD (<= 256 VGPRs, no spills)
#ifndef PHMM_SDWA_MIN_K
#define PHMM_SDWA_MIN_K 21
#endif
// K up to this value also gets the v_cndmask body for launches that leave a wave alone on its SIMD
#ifndef PHMM_CND_MAX_K
#define PHMM_CND_MAX_K 13
#endif
#ifndef PHMM_TWO_WAVE_MAX_K
#define PHMM_TWO_WAVE_MAX_K 25
#endif

namespace phmm {

// ---- DPP lane shifts (zero fill where there is no source lane) ---------------------------------
__device__ __forceinline__ int dpp_row_shr1(int v) {  // lane n <- lane n-1 inside each row of 16
    return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
}
__device__ __forceinline__ int dpp_wave_shr1(int v) {  // lane n <- lane n-1 across the whole wave
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}

template <int L>
__device__ __forceinline__ double from_left(double v, bool group_head) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (L == 16) {
        lo = dpp_row_shr1(lo);
        hi = dpp_row_shr1(hi);
    } else {
        lo = dpp_wave_shr1(lo);
        hi = dpp_wave_shr1(hi);
        if constexpr (L == 32) {  // lane 32 starts the second pair: its left boundary is column 0
            lo = group_head ? 0 : lo;
            hi = group_head ? 0 : hi;
        }
    }
    return __hiloint2double(hi, lo);
}

// Per read row, staged in wave-private LDS as one 72-byte record: all fields of a row are reached
// from ONE address register with immediate offsets.  A 72-byte stride (18 dwords) maps 32 consecutive
// rows onto 32 distinct bank pairs, so the staggered per-lane reads (lane l reads row t-l) are
// conflict-free, and lanes of different haplotype groups reading the same row broadcast.
//
// The record holds the coefficients of the row update in the form the kernel evaluates it:
//   M~(i,k) = sel * ( M~(i-1,k-1)*mm + (I^(i-1,k-1) + D'(i-1,k-1)*dDp) * imx ),  sel = x==y ? pm : px
//   I^(i,k) = M~(i-1,k)*bI + I^(i-1,k)*gI
//   D'(i,k) = M~(i,k-1)    + D'(i,k-1)*dd          (one FMA: the row's M->D factor is applied by the consumer)
// Plain rows (any read; used when a read has gcp == 0 or a base quality 0):
//   M~ = M, I^ = I, D' = D/md(i); mm, bI = mi, gI = ii, dd = ii as the reference writes them, dDp = md(i-1)
//   (1 for the first row: D'(0,.) = D(0,.)), pm/px = the match / mismatch priors, imx = im = 1 - ii.
// Pre-scaled rows (everything else): three factors are folded into neighbouring coefficients so that the
// steady-state body is 7 f64-rate VALU ops per cell (fma, fma, masked mul, mul, fma, fma, compare):
//   * I^(i) = I(i)*im(i+1), D^(i) = D(i)*im(i+1) with im(R+1) = 1: the indel->match factor (imx == 1):
//       bI = mi*im(i+1), gI = ii*im(i+1)/im(i), dD = md*im(i+1)
//   * D'(i) = D^(i)/dD(i): the match->deletion factor moves to the consumer (dDp = dD(i-1))
//   * M~(i) = M(i)/pm(i): a matching cell needs no prior multiply at all (pm field = 1.0) and a mismatching
//       cell is multiplied by px = prior_mismatch/prior_match; pm(i-1) is folded into mm, bI and dDp of row i.
//       The final sum is sum_j M~(R,j)*pm(R) + I^(R,j).
struct alignas(8) RowConst {
    double mm, bI, gI, dDp, dd, pm, px;  // see above
    uint32_t x, pad0;                   // read base
    double pad1;
};
static_assert(sizeof(RowConst) == 72, "LDS row record");

// Record `idx` of an LDS array of row records.  The byte offset is formed with a 24-bit multiply-add (one 32-bit
// op) instead of the 64-bit v_mad_u64_u32 / v_mul_lo_u32 the compiler picks for a plain index; indices are far
// below 2^24.  (Measured: no difference on gfx950 -- kept because it is the cheaper encoding.)
__device__ __forceinline__ RowConst lds_row(const RowConst *rows, int idx) {
    return *reinterpret_cast<const RowConst *>(reinterpret_cast<const unsigned char *>(rows) +
                                               __mul24(idx, (int)sizeof(RowConst)));
}

// The record at LDS byte address `at` (+ `skip` records): nine 64-bit LDS reads off one address register, the record
// offset in their immediate fields.
__device__ __forceinline__ RowConst lds_row_at(uint32_t at, int skip) {
    RowConst r;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(3))) unsigned long long *LdsU64;
    const LdsU64 w = (LdsU64)(uintptr_t)at + skip * (int)(sizeof(RowConst) / 8);
    unsigned long long v[sizeof(RowConst) / 8];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(RowConst) / 8) - 1; ++i) v[i] = w[i];
    v[8] = w[8];
    __builtin_memcpy(&r, v, sizeof r);
#else
    (void)at;
    (void)skip;
    r = RowConst{};
#endif
    return r;
}

struct LdsView {
    const RowConst *rows;  // index 0 = neutral row, read row r at index r+1
    __device__ __forceinline__ RowConst load(int idx) const { return lds_row(rows, idx); }
};

// compile-time k = K-1 .. 0
template <int K, class F>
__device__ __forceinline__ void static_for_down(F &&f) {
    if constexpr (K > 0) {
        f(std::integral_constant<int, K - 1>{});
        static_for_down<K - 1>(f);
    }
}

// Haplotype columns of a lane, two 16-bit fields per dword (column k of the lane = l*K+k): the fast body compares
// straight out of the packed half-word (SDWA operand select), so a column costs half a register.
template <int K>
struct HapCols {
    static constexpr int W = (K + 1) / 2;
    uint32_t y[W];  // raw haplotype byte (or a chained kernel's EDGE / PAD code)
    __device__ __forceinline__ uint16_t base(int k) const { return (uint16_t)(y[k >> 1] >> (16 * (k & 1))); }
    __device__ __forceinline__ void set(int k, uint32_t yv) { y[k >> 1] |= yv << (16 * (k & 1)); }
};

// One read row for the K columns of this lane, every register updated in place.
//   in : Mp/Ip/Dp = row i-1;  (plM,plI,plD) = left neighbour's last column, row i-1;
//        (lM,lD) = left neighbour's last column, row i
//   out: Mp/Ip/Dp = row i
// FAST_EXEC: pre-scaled rows and a haplotype without 'N' (the common case), every lane of the wave active;
// FAST_EXEC_PRED: the same under a predicate (`live` = the EXEC mask the caller runs under, wave-uniform);
// FAST_CND: the same arithmetic with a v_cndmask select, scheduled by the compiler -- one VALU op more per cell but
// no EXEC round trip, which is faster when a wave has a SIMD to itself (small launches; compiled for small K only).
// Otherwise the general form: `imx` multiplies the indel->match term (1.0 for pre-scaled rows) and the
// compare honours the haplotype wildcard mask.
enum : int { ROW_GENERAL = 0, ROW_FAST_CND = 1, ROW_FAST_EXEC = 2, ROW_FAST_EXEC_PRED = 3 };

template <int K, int MODE>
__device__ __forceinline__ void row_update(double (&Mp)[K], double (&Ip)[K], double (&Dp)[K], const double plM,
                                           const double plI, const double plD, const double lM, const double lD,
                                           const RowConst &c, const HapCols<K> &hc, const double imx,
                                           const uint64_t live = ~0ull) {
    const uint16_t x16 = (uint16_t)c.x;
    if constexpr (MODE == ROW_FAST_EXEC || MODE == ROW_FAST_EXEC_PRED) {
        // Pre-scaled rows (pm == 1).  One cell = one asm statement in a fixed order:
        //   M~(k)  = D'(k-1)*dDp + I^(k-1);  M~(k) += M~(k-1)*mm          (row i-1 values of column k-1)
        //   EXEC = mask_k;  M~(k) *= px;  EXEC = live                     (mask_k = lanes where x != y_k: matching
        //                                                                  cells keep the value)
        //   mask_(k-1) = (x != y_(k-1))                                   (v_cmp into an SGPR pair, one cell ahead)
        //   I^(k-1) = I^(k-1)*gI + M~(k-1)*bI                             (column k-1 moves on to row i)
        // Columns right-to-left, so every register is updated in place; I^(K-1) and mask_(K-1) come first.
        // The mismatch mask is made by a plain v_cmp one cell AHEAD of its use and moved into EXEC by the scalar
        // unit: a VALU instruction that writes EXEC itself (v_cmpx) stalls the SIMD for the round trip -- 4.8 vs 3.9
        // clk per instruction for this very cell at two waves per SIMD (tools/ubench/issue.hip).  The compare runs
        // under EXEC = live, so the mask never has a bit outside it.
        // The accumulating FMAs are written in the VOP3 form on purpose: the 2-address VOP2 v_fmac_f64_e32 issues
        // at ~60 % of the VOP3 rate on gfx950 (tools/ubench/banks.hip).
        const uint64_t restore = (MODE == ROW_FAST_EXEC) ? ~0ull : live;
        // K > PHMM_SDWA_MIN_K: the haplotype base is compared straight out of its packed half-word (SDWA operand
        // select) so that no unpacked copy per column lives in registers -- that is what lets K = 22..25 keep
        // two waves per SIMD; smaller K can afford the copies and the plain compare issues a little faster.
        constexpr bool packed = K > PHMM_SDWA_MIN_K;
        constexpr bool pred = MODE == ROW_FAST_EXEC_PRED;
        uint64_t mask;  // mismatch mask of the column about to be updated
        {
            constexpr int k = K - 1;
            const uint32_t y = packed ? hc.y[k >> 1] : (uint32_t)hc.base(k);
            if constexpr (packed && (k & 1))
                asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:DWORD src1_sel:WORD_1" : "=s"(mask) : "v"(c.x), "v"(y));
            else if constexpr (packed)
                asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:DWORD src1_sel:WORD_0" : "=s"(mask) : "v"(c.x), "v"(y));
            else
                asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(mask) : "v"(c.x), "v"(y));
        }
        Ip[K - 1] = fma(Mp[K - 1], c.bI, Ip[K - 1] * c.gI);
        static_for_down<K>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int km1 = k > 0 ? k - 1 : 0;
            const uint32_t yn = packed ? hc.y[km1 >> 1] : (uint32_t)hc.base(km1);  // the NEXT column to be updated
            uint64_t next;
#define PHMM_CELL(CMP, RESTORE)                                                                                      \
    asm volatile("v_fma_f64 %[M], %[Dl], %[dDp], %[Il]\n\t"                                                        \
                 "v_fma_f64 %[M], %[Ml], %[mm], %[M]\n\t"                                                          \
                 "s_mov_b64 exec, %[mk]\n\t"                                                                       \
                 "v_mul_f64 %[M], %[px], %[M]\n\t"                                                                 \
                 "s_mov_b64 exec, " RESTORE "\n\t"                                                                 \
                 CMP "\n\t"                                                                                      \
                 "v_mul_f64 %[Il], %[Il], %[gI]\n\t"                                                               \
                 "v_fma_f64 %[Il], %[Ml], %[bI], %[Il]"                                                             \
                 : [M] "=&v"(Mp[k]), [Il] "+v"(Ip[km1]), [mn] "=&s"(next)                                           \
                 : [Dl] "v"(Dp[km1]), [dDp] "v"(c.dDp), [Ml] "v"(Mp[km1]), [mm] "v"(c.mm), [x] "v"(c.x), [y] "v"(yn), \
                   [px] "v"(c.px), [gI] "v"(c.gI), [bI] "v"(c.bI), [live] "s"(restore), [mk] "s"(mask))
            if constexpr (k > 0 && pred && packed && (km1 & 1)) {
                PHMM_CELL("v_cmp_ne_u32_sdwa %[mn], %[x], %[y] src0_sel:DWORD src1_sel:WORD_1", "%[live]");
            } else if constexpr (k > 0 && pred && packed) {
                PHMM_CELL("v_cmp_ne_u32_sdwa %[mn], %[x], %[y] src0_sel:DWORD src1_sel:WORD_0", "%[live]");
            } else if constexpr (k > 0 && pred) {
                PHMM_CELL("v_cmp_ne_u32_e64 %[mn], %[x], %[y]", "%[live]");
            } else if constexpr (k > 0 && packed && (km1 & 1)) {
                PHMM_CELL("v_cmp_ne_u32_sdwa %[mn], %[x], %[y] src0_sel:DWORD src1_sel:WORD_1", "-1");
            } else if constexpr (k > 0 && packed) {
                PHMM_CELL("v_cmp_ne_u32_sdwa %[mn], %[x], %[y] src0_sel:DWORD src1_sel:WORD_0", "-1");
            } else if constexpr (k > 0) {
                PHMM_CELL("v_cmp_ne_u32_e64 %[mn], %[x], %[y]", "-1");
#undef PHMM_CELL
            } else {
                (void)yn;
                double m = fma(plM, c.mm, fma(plD, c.dDp, plI));
                asm volatile("s_mov_b64 exec, %2\n\t"
                             "v_mul_f64 %0, %1, %0\n\t"
                             "s_mov_b64 exec, %3"
                             : "+v"(m)
                             : "v"(c.px), "s"(mask), "s"(restore));
                Mp[0] = m;
                next = 0;
            }
            mask = next;
        });
        double leftM = lM, leftD = lD;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            Dp[k] = fma(leftD, c.dd, leftM);
            leftM = Mp[k];
            leftD = Dp[k];
        }
        return;
    }
    // General form (rare).  Pass 1, columns right-to-left: I(i,k) reads the old M/I of column k, then M(i,k) overwrites
    // M[k] using the still-old column k-1.
    static_for_down<K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // written so the two-address FMA accumulates into I's own register (v_mul I,I,gI ; v_fmac I,M,bI)
        Ip[k] = fma(Mp[k], c.bI, Ip[k] * c.gI);
        constexpr int km1 = k > 0 ? k - 1 : 0;
        const double dM = k > 0 ? Mp[km1] : plM;  // (i-1, k-1)
        const double dI = k > 0 ? Ip[km1] : plI;
        const double dD = k > 0 ? Dp[km1] : plD;
        double t = fma(dD, c.dDp, dI);  // I^(i-1,k-1) + D^(i-1,k-1), with D^ = D' * dD(i-1)
        if constexpr (MODE == ROW_GENERAL) t *= imx;
        const double a = fma(dM, c.mm, t);
        const uint16_t yb = hc.base(k);  // haplotype 'N' is a wildcard too (pair_hmm.rs:643)
        const double prior = (x16 == yb || (MODE == ROW_GENERAL && yb == (uint16_t)'N')) ? c.pm : c.px;
        Mp[k] = prior * a;
    });
    // Pass 2, left-to-right: the serial chain D'(i,k) = M(i,k-1) + D'(i,k-1)*dd.
    double leftM = lM, leftD = lD;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Dp[k] = fma(leftD, c.dd, leftM);
        leftM = Mp[k];
        leftD = Dp[k];
    }
}

// Sweeps.  Step t: lane l works on read row t-l.  LDS row index 0 holds a NEUTRAL row (bI=dD=0,
// gI=dd=1, pm=px=0) under which the row-0 state (M=0, I=0, D=c0) is an exact fixed point, so lanes
// that have not started yet simply run it: the first R steps need no predicate at all.  Only the L-1
// drain steps (lanes past their last row must freeze) are predicated.
// c0 = D(0,j) = 2^1020/H (pair_hmm.rs:515-529), times im(1) for pre-scaled rows.

// Fast sweep: two steps per trip with the roles of the (constants, left-column) register sets swapped,
// so nothing is copied between steps.  Returns this lane's partial of sum_j M[R][j]+I[R][j].
template <int L, int K, int STEADY = ROW_FAST_EXEC>
__device__ __forceinline__ double sweep_fast(const LdsView &lds, const int R, const int l, const bool group_head,
                                             const HapCols<K> &hc, const int H, const double c0, const double fin) {
    double Mp[K], Ip[K], Dp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Mp[k] = 0.0;
        Ip[k] = 0.0;
        Dp[k] = c0;
    }
    double aM, aI, aD, bM = 0.0, bI = 0.0, bD = c0;  // left neighbour's last column: row i / row i-1
    int row = -l;  // 0-based read row of this lane at step t (= t - l); LDS index = row + 1
    RowConst cA = lds.load(max(row + 1, 0)), cB;
    int t = 0;
    for (; t + 1 < R; t += 2) {  // fill + steady state: no lane has finished yet, no predicate
        cB = lds.load(max(row + 2, 0));  // one step ahead (index R at most)
        aM = from_left<L>(Mp[K - 1], group_head);
        aI = from_left<L>(Ip[K - 1], group_head);
        aD = from_left<L>(Dp[K - 1], group_head);
        row_update<K, STEADY>(Mp, Ip, Dp, bM, bI, bD, aM, aD, cA, hc, 1.0);
        cA = lds.load(max(row + 3, 0));
        bM = from_left<L>(Mp[K - 1], group_head);
        bI = from_left<L>(Ip[K - 1], group_head);
        bD = from_left<L>(Dp[K - 1], group_head);
        row_update<K, STEADY>(Mp, Ip, Dp, aM, aI, aD, bM, bD, cB, hc, 1.0);
        row += 2;
    }
    RowConst cur = cA;
    double plM = bM, plI = bI, plD = bD;
    if (t < R) {  // odd read length: one more unpredicated step
        const RowConst nxt = lds.load(max(row + 2, 0));
        const double lM = from_left<L>(Mp[K - 1], group_head);
        const double lI = from_left<L>(Ip[K - 1], group_head);
        const double lD = from_left<L>(Dp[K - 1], group_head);
        row_update<K, STEADY>(Mp, Ip, Dp, plM, plI, plD, lM, lD, cur, hc, 1.0);
        plM = lM;
        plI = lI;
        plD = lD;
        ++row;
        cur = nxt;
    }
    for (int d = 0; d < L - 1; ++d) {  // drain: lane l still has rows while row < R
        const RowConst nxt = lds.load(max(min(row + 2, R), 0));
        const double lM = from_left<L>(Mp[K - 1], group_head);
        const double lI = from_left<L>(Ip[K - 1], group_head);
        const double lD = from_left<L>(Dp[K - 1], group_head);
        const uint64_t live = __ballot(row < R);
        if (row < R)
            row_update<K, STEADY == ROW_FAST_CND ? ROW_FAST_CND : ROW_FAST_EXEC_PRED>(Mp, Ip, Dp, plM, plI, plD, lM, lD, cur,
                                                                                     hc, 1.0, live);
        plM = lM;
        plI = lI;
        plD = lD;
        ++row;
        cur = nxt;
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (l * K + k < H) s += fma(Mp[k], fin, Ip[k]);  // fin = pm(R) for pre-scaled rows, else 1
    return s;
}

// General sweep (haplotype with 'N', or a read with gcp == 0 whose rows cannot be pre-scaled):
// one compact predicated loop, kept small on purpose -- it is rare.
template <int L, int K, class RowView>
__device__ __forceinline__ double sweep_general(const RowView &lds, const int R, const int l, const bool group_head,
                                             const HapCols<K> &hc, const int H, const double c0, const bool scaled,
                                             const double fin) {
    double Mp[K], Ip[K], Dp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Mp[k] = 0.0;
        Ip[k] = 0.0;
        Dp[k] = c0;
    }
    double plM = 0.0, plI = 0.0, plD = c0;
    int row = -l;
    for (int t = 0; t < R + L - 1; ++t) {
        const RowConst cur = lds.load(max(min(row + 1, R), 0));
        const double lM = from_left<L>(Mp[K - 1], group_head);
        const double lI = from_left<L>(Ip[K - 1], group_head);
        const double lD = from_left<L>(Dp[K - 1], group_head);
        const double imx = scaled ? 1.0 : 1.0 - cur.dd;  // plain rows: dd == ii, im = 1 - ii
        if (row < R) row_update<K, ROW_GENERAL>(Mp, Ip, Dp, plM, plI, plD, lM, lD, cur, hc, imx);
        plM = lM;
        plI = lI;
        plD = lD;
        ++row;
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (l * K + k < H) s += fma(Mp[k], fin, Ip[k]);  // fin = pm(R) for pre-scaled rows, else 1
    return s;
}


// One row record from the read's quality bytes (tables live in HBM / L2); forms as described at RowConst.
// `q_prev` / `dq_prev` belong to the previous row (ignored when `first`), `g_next` to the following row
// (ignored when `last`).
__device__ __forceinline__ RowConst make_row_bytes(const ForwardParams &p, uint32_t x, uint32_t q, uint32_t q_prev,
                                                   uint32_t iq, uint32_t dq, uint32_t dq_prev, uint32_t g,
                                                   uint32_t g_next, bool first, bool last, bool scaled) {
    const double mi = p.eps[iq], ii = p.eps[g];
    const uint32_t mx = max(iq, dq), mn = min(iq, dq);
    const double mm = p.mm[((mx * (mx + 1)) >> 1) + mn];  // pair_hmm_model.rs:442-461
    RowConst n;
    n.dd = ii;
    if (scaled) {  // division-free: the quotients come from host-built tables
        const double im = 1.0 - ii;
        const double im_next = last ? 1.0 : 1.0 - p.eps[g_next];
        const double pm_prev = first ? 1.0 : 1.0 - p.eps[q_prev];
        n.mm = mm * pm_prev;
        n.bI = mi * im_next * pm_prev;
        n.gI = ii * (im_next * p.inv_om[g]);
        n.dDp = first ? 1.0 : p.eps[dq_prev] * im * pm_prev;  // dD(i-1) = md(i-1) * im(i)
        n.pm = 1.0;
        n.px = (x == 'N') ? 1.0 : p.ratio_mis[q];  // read 'N' matches everything (pair_hmm.rs:643)
    } else {
        const double pm = 1.0 - p.eps[q];                  // qual_to_prob(q)
        const double px = (x == 'N') ? pm : p.eps_mis[q];
        n.mm = mm;
        n.bI = mi;
        n.gI = ii;
        n.dDp = first ? 1.0 : p.eps[dq_prev];
        n.pm = pm;
        n.px = px;
    }
    n.x = x;
    n.pad0 = 0;
    n.pad1 = 0.0;
    return n;
}

__device__ __forceinline__ RowConst make_row(const ForwardParams &p, uint32_t ro, int row, int R, bool scaled) {
    const bool first = row == 0, last = row + 1 >= R;
    return make_row_bytes(p, p.read_bases[ro + row], p.base_q[ro + row], first ? 0u : (uint32_t)p.base_q[ro + row - 1],
                          p.ins_q[ro + row], p.del_q[ro + row], first ? 0u : (uint32_t)p.del_q[ro + row - 1],
                          p.gcp[ro + row], last ? 0u : (uint32_t)p.gcp[ro + row + 1], first, last, scaled);
}

__device__ __forceinline__ RowConst neutral_row();

// Row source that builds each record from HBM on the fly (no staging): for the rare general path of the chained
// kernel when a read does not fit its LDS ring.  Same indexing as LdsView (0 = neutral row).
struct GlobalRowView {
    const ForwardParams &p;
    uint32_t ro;
    int R;
    bool scaled;
    __device__ __forceinline__ RowConst load(int idx) const {
        return idx == 0 ? neutral_row() : make_row(p, ro, idx - 1, R, scaled);
    }
};

// A read can use pre-scaled rows unless one of its rows has gcp == 0 (im = 0) or base quality 0 (pm = 0).
__device__ __forceinline__ bool row_blocks_prescale(const ForwardParams &p, uint32_t byte) {
    return p.gcp[byte] == 0 || p.base_q[byte] == 0;
}

__device__ __forceinline__ RowConst neutral_row() {  // keeps (M, I^, D') = (0, 0, c0) fixed
    RowConst n;
    n.mm = 0.0; n.bI = 0.0; n.gI = 1.0; n.dDp = 0.0; n.dd = 1.0; n.pm = 0.0; n.px = 0.0;
    n.x = 0; n.pad0 = 0; n.pad1 = 0.0;
    return n;
}

__device__ __forceinline__ void lds_wave_sync() {
    // LDS ops of one wave execute in order; only the compiler must not reorder across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One read against the haplotype groups quad_begin, quad_begin + quad_step, ... of its region: the body of phmm_forward<L,K>
// (phmm_kernels.hip) and of the resident region server's forward task (phmm_server_kernels.hip).  `smem_wave`: this wave's
// own LDS, (R + 1) row records.  Wave-uniform control flow; every lane of the wave must arrive.
template <int L, int K>
__device__ __forceinline__ void forward_read(const ForwardParams &p, const uint32_t r, const int quad_begin, const int quad_step,
                                             const bool cnd_select, unsigned char *smem_wave, double *coherent_out = nullptr) {
    constexpr int G = WAVE / L;
    const int lane = threadIdx.x & (WAVE - 1);
    const int grp = lane / L, l = lane % L;
    const uint32_t reg = p.read_region[r];
    const uint32_t ro = p.read_off[r];
    const int R = (int)(p.read_off[r + 1] - ro);
    const uint32_t h0 = p.region_hap_off[reg];
    const int Nh = (int)(p.region_hap_off[reg + 1] - h0);
    double *out_row = p.out + p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh;

    // ---- stage this read's per-row constants in wave-private LDS (72-byte records, conflict-free) ----------
    RowConst *srow = reinterpret_cast<RowConst *>(smem_wave);
    // gcp == 0 means im = 1 - eps(0) = 0 (and base quality 0 means pm = 0): such a read keeps plain rows
    // (rare; production gcp is 10 and the engine caps base qualities at >= 6)
    bool lane_zero_gcp = false;
    for (int row = lane; row < R; row += WAVE) lane_zero_gcp |= row_blocks_prescale(p, ro + row);
    const bool scaled = __ballot(lane_zero_gcp) == 0ull;
    if (lane == 0) srow[0] = neutral_row();  // lanes that have not started yet run this row
    for (int row = lane; row < R; row += WAVE) srow[row + 1] = make_row(p, ro, row, R, scaled);
    // D(0,j) scale: pre-scaled rows carry im of the first read row
    const double scale0 = (scaled && R > 0) ? 1.0 - p.eps[p.gcp[ro]] : 1.0;
    const double fin = (scaled && R > 0) ? 1.0 - p.eps[p.base_q[ro + R - 1]] : 1.0;  // pm(R), see RowConst
    lds_wave_sync();
    const LdsView lds{srow};
    const bool group_head = (L == 32) && (lane == 32);

    const int nquads = (Nh + G - 1) / G;
    for (int quad = quad_begin; quad < nquads; quad += quad_step) {
        const int a = quad * G + grp;
        const bool hv = a < Nh;
        uint32_t ho = 0;
        int H = 0;
        if (hv) {
            ho = p.hap_off[h0 + a];
            H = (int)(p.hap_off[h0 + a + 1] - ho);
        }
        HapCols<K> hc;
        bool lane_n = false;
#pragma unroll
        for (int w = 0; w < HapCols<K>::W; ++w) hc.y[w] = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int col = l * K + k;
            const uint32_t y = (col < H) ? (uint32_t)p.hap_bases[ho + col] : 0u;
            const bool is_n = (y == 'N');
            lane_n |= is_n;
            hc.set(k, y);
        }
        const double c0 = p.initial_condition / (double)H * scale0;
        double s;
        if (scaled && __ballot(lane_n) == 0ull) {
            if constexpr (K <= PHMM_CND_MAX_K) {
                s = cnd_select ? sweep_fast<L, K, ROW_FAST_CND>(lds, R, l, group_head, hc, H, c0, fin)
                               : sweep_fast<L, K>(lds, R, l, group_head, hc, H, c0, fin);
            } else {
                s = sweep_fast<L, K>(lds, R, l, group_head, hc, H, c0, fin);
            }
        } else {  // rare: haplotype 'N' is a wildcard too (pair_hmm.rs:643), or a read with gcp == 0 / base quality 0
            s = sweep_general<L, K>(lds, R, l, group_head, hc, H, c0, scaled, fin);
        }
#pragma unroll
        for (int off = L / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        if (l == 0 && hv) {
            const double v = log10(s) - p.initial_condition_log10;
            // (coherent_out: a helper wave of the region server, whose results a wave of another XCD reads -- agent-scope stores
            // into words that only such accesses ever touch)
            if (coherent_out) __hip_atomic_store(&coherent_out[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else out_row[a] = v;
            if (const uint32_t sb = status_bits(v)) atomicOr(p.status, sb);  // reference asserts result <= 0 (pair_hmm.rs:478-481)
        }
    }
}

}  // namespace phmm
