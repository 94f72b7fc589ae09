// gfx950 (CDNA4, wave64) kernels of the PairHMM forward path.
//
// Recurrence (reference: src/pair_hmm/pair_hmm.rs:503-615, priors :626-673, transitions
// src/pair_hmm/pair_hmm_model.rs:142-156), all in linear space, f64, scaled by 2^1020:
//   M[i][j] = prior(i,j) * ( M[i-1][j-1]*mm_i + I[i-1][j-1]*im_i + D[i-1][j-1]*im_i )
//   I[i][j] = M[i-1][j]*mi_i + I[i-1][j]*ii_i
//   D[i][j] = M[i][j-1]*md_i + D[i][j-1]*dd_i
//   result  = log10( sum_j M[R][j] + I[R][j] ) - log10(2^1020)
//
// Mapping (see DESIGN.md "Kernel"): the HAPLOTYPE runs along the lanes.  A group of L lanes owns
// one (read, haplotype) pair; lane l of the group keeps K consecutive haplotype columns
// (j = l*K+1 .. l*K+K) of the previous read row in registers (3*K f64) and walks down the read one
// row per step, one step behind lane l-1 -- i.e. the wave sweeps anti-diagonals of K-column
// blocks.  The only cross-lane traffic per step is the last column of the left neighbour
// (M, I, D = 6 dwords) moved with DPP row_shr:1 / wave_shr:1; the per-row transition / prior
// constants are staged once per read in wave-private LDS and fetched one step ahead.
// 64/L pairs share a wave (same read, different haplotypes), so all groups have the same trip
// count and there is no divergence except the start-up / drain predicate.
#include "phmm_internal.hpp"

#include <type_traits>

// K up to this value is compiled for two resident waves per SIMD (<= 256 VGPRs, no spills)
#ifndef PHMM_TWO_WAVE_MAX_K
#define PHMM_TWO_WAVE_MAX_K 21
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
//   M(i,k) = prior * ( M(i-1,k-1)*mm + (I^(i-1,k-1) + D^(i-1,k-1)) * imx )
//   I^(i,k) = M(i-1,k)*bI + I^(i-1,k)*gI
//   D^(i,k) = M(i,k-1)*dD + D^(i,k-1)*dd
// Plain rows (any read):       I^ = I, D^ = D, bI = mi, gI = ii, dD = md, dd = ii, imx = im = 1 - dd.
// Pre-scaled rows (no gcp==0): I^(i) = I(i)*im(i+1), D^(i) = D(i)*im(i+1) with im(R+1) = 1, so the
//   indel->match factor is already folded in (imx == 1, one f64 op less per cell):
//   bI = mi*im(i+1), gI = ii*im(i+1)/im(i), dD = md*im(i+1), dd = ii.
struct alignas(8) RowConst {
    double mm, bI, gI, dD, dd, pm, px;  // pm = 1 - eps(q) (match prior), px = mismatch prior
    uint32_t x, pad0;                   // read base
    double pad1;
};
static_assert(sizeof(RowConst) == 72, "LDS row record");

struct LdsView {
    const RowConst *rows;  // index 0 = neutral row, read row r at index r+1
    __device__ __forceinline__ RowConst load(int idx) const { return rows[idx]; }
};

// compile-time k = K-1 .. 0
template <int K, class F>
__device__ __forceinline__ void static_for_down(F &&f) {
    if constexpr (K > 0) {
        f(std::integral_constant<int, K - 1>{});
        static_for_down<K - 1>(f);
    }
}

// Haplotype columns of a lane, two 16-bit fields per dword (column k of the lane = l*K+k): the
// compare then is a single v_cmp_eq_u16 with a half-word select, no extraction ops.
template <int K>
struct HapCols {
    static constexpr int W = (K + 1) / 2;
    uint32_t y[W];  // base (0 where the haplotype has 'N' and HAPN is set)
    uint32_t m[W];  // HAPN only: 0xff = compare, 0x00 = wildcard column
    __device__ __forceinline__ uint16_t base(int k) const { return (uint16_t)(y[k >> 1] >> (16 * (k & 1))); }
    __device__ __forceinline__ uint16_t mask(int k) const { return (uint16_t)(m[k >> 1] >> (16 * (k & 1))); }
    __device__ __forceinline__ void set(int k, uint32_t yv, uint32_t mv) {
        y[k >> 1] |= yv << (16 * (k & 1));
        m[k >> 1] |= mv << (16 * (k & 1));
    }
};

// One read row for the K columns of this lane, every register updated in place.
//   in : Mp/Ip/Dp = row i-1;  (plM,plI,plD) = left neighbour's last column, row i-1;
//        (lM,lD) = left neighbour's last column, row i
//   out: Mp/Ip/Dp = row i
// FAST: pre-scaled rows and a haplotype without 'N' (the common case).  Otherwise the general form:
// `imx` multiplies the indel->match term (1.0 for pre-scaled rows) and the compare honours the
// haplotype wildcard mask.
enum : int { ROW_GENERAL = 0, ROW_FAST = 1, ROW_FAST_EXEC = 2 };

template <int K, int MODE>
__device__ __forceinline__ void row_update(double (&Mp)[K], double (&Ip)[K], double (&Dp)[K], const double plM,
                                           const double plI, const double plD, const double lM, const double lD,
                                           const RowConst &c, const HapCols<K> &hc, const double imx) {
    const uint16_t x16 = (uint16_t)c.x;
    // Pass 1, columns right-to-left: I(i,k) reads the old M/I of column k, then M(i,k) overwrites
    // M[k] using the still-old column k-1.
    static_for_down<K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // written so the two-address FMA accumulates into I's own register (v_mul I,I,gI ; v_fmac I,M,bI)
        Ip[k] = fma(Mp[k], c.bI, Ip[k] * c.gI);
        constexpr int km1 = k > 0 ? k - 1 : 0;
        const double dM = k > 0 ? Mp[km1] : plM;  // (i-1, k-1)
        const double dI = k > 0 ? Ip[km1] : plI;
        const double dD = k > 0 ? Dp[km1] : plD;
        double t = dI + dD;
        if constexpr (MODE == ROW_GENERAL) t *= imx;
        const double a = fma(dM, c.mm, t);
        if constexpr (MODE == ROW_FAST_EXEC) {
            // prior select without v_cndmask: multiply by the mismatch prior everywhere, then redo the
            // multiply with the match prior under EXEC = (x == y).  Two VALU + one SALU instead of four
            // VALU (compare, two v_cndmask, multiply).  Only valid where all 64 lanes are active.
            double m = c.px * a;
            asm volatile("v_cmpx_eq_u32_e32 vcc, %1, %2\n\t"
                         "v_mul_f64 %0, %3, %4\n\t"
                         "s_mov_b64 exec, -1"
                         : "+v"(m)
                         : "v"(c.x), "v"((uint32_t)hc.base(k)), "v"(c.pm), "v"(a)
                         : "vcc");
            Mp[k] = m;
        } else {
            double prior;
            if constexpr (MODE == ROW_FAST)
                prior = (x16 == hc.base(k)) ? c.pm : c.px;
            else
                prior = ((uint16_t)(x16 & hc.mask(k)) == hc.base(k)) ? c.pm : c.px;
            Mp[k] = prior * a;
        }
    });
    // Pass 2, left-to-right: the serial chain D(i,k) = M(i,k-1)*dD + D(i,k-1)*dd.
    double leftM = lM, leftD = lD;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Dp[k] = fma(leftD, c.dd, leftM * c.dD);
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
template <int L, int K, int STEADY>
__device__ __forceinline__ double sweep_fast(const LdsView &lds, const int R, const int l, const bool group_head,
                                             const HapCols<K> &hc, const int H, const double c0) {
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
        if (row < R) row_update<K, ROW_FAST>(Mp, Ip, Dp, plM, plI, plD, lM, lD, cur, hc, 1.0);
        plM = lM;
        plI = lI;
        plD = lD;
        ++row;
        cur = nxt;
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (l * K + k < H) s += Mp[k] + Ip[k];
    return s;
}

// General sweep (haplotype with 'N', or a read with gcp == 0 whose rows cannot be pre-scaled):
// one compact predicated loop, kept small on purpose -- it is rare.
template <int L, int K>
__device__ __forceinline__ double sweep_general(const LdsView &lds, const int R, const int l, const bool group_head,
                                             const HapCols<K> &hc, const int H, const double c0, const bool scaled) {
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
        if (l * K + k < H) s += Mp[k] + Ip[k];
    return s;
}

template <int L, int K>
__global__ __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK, (K <= PHMM_TWO_WAVE_MAX_K ? 2 : 1)) void phmm_forward(const ForwardParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int G = WAVE / L;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane / L, l = lane % L;
    const uint32_t item = blockIdx.x * (blockDim.x >> 6) + wave;
    if (item >= p.n_items) return;  // wave-uniform
    const uint32_t r = p.class_reads ? p.class_reads[item] : item;
    const uint32_t reg = p.read_region[r];
    const uint32_t ro = p.read_off[r];
    const int R = (int)(p.read_off[r + 1] - ro);
    const uint32_t h0 = p.region_hap_off[reg];
    const int Nh = (int)(p.region_hap_off[reg + 1] - h0);
    double *out_row = p.out + p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh;

    // ---- stage this read's per-row constants in wave-private LDS (72-byte records, conflict-free) ----------
    RowConst *srow = reinterpret_cast<RowConst *>(smem) + (size_t)wave * p.lds_rows;
    // gcp == 0 means im = 1 - eps(0) = 0: such a read keeps plain rows (rare; production gcp is 10)
    bool lane_zero_gcp = false;
    for (int row = lane; row < R; row += WAVE) lane_zero_gcp |= (p.gcp[ro + row] == 0);
    const bool scaled = __ballot(lane_zero_gcp) == 0ull;
    if (lane == 0) {  // neutral row: keeps (M, I, D) = (0, 0, c0) fixed for lanes that have not started
        RowConst n;
        n.mm = 0.0; n.bI = 0.0; n.gI = 1.0; n.dD = 0.0; n.dd = 1.0; n.pm = 0.0; n.px = 0.0;
        n.x = 0; n.pad0 = 0; n.pad1 = 0.0;
        srow[0] = n;
    }
    for (int row = lane; row < R; row += WAVE) {
        const uint32_t x = p.read_bases[ro + row];
        const uint32_t q = p.base_q[ro + row];
        const uint32_t iq = p.ins_q[ro + row];
        const uint32_t dq = p.del_q[ro + row];
        const uint32_t g = p.gcp[ro + row];
        const uint32_t mx = max(iq, dq), mn = min(iq, dq);
        const double eq = p.eps[q], mi = p.eps[iq], md = p.eps[dq], ii = p.eps[g];
        RowConst n;
        n.mm = p.mm[((mx * (mx + 1)) >> 1) + mn];  // pair_hmm_model.rs:442-461
        n.pm = 1.0 - eq;                           // qual_to_prob(q)
        n.px = (x == 'N') ? n.pm : p.eps_mis[q];   // read 'N' matches everything (pair_hmm.rs:643)
        n.dd = ii;
        if (scaled) {
            const double im = 1.0 - ii;
            const double im_next = (row + 1 < R) ? 1.0 - p.eps[p.gcp[ro + row + 1]] : 1.0;
            n.bI = mi * im_next;
            n.gI = ii * (im_next / im);
            n.dD = md * im_next;
        } else {
            n.bI = mi;
            n.gI = ii;
            n.dD = md;
        }
        n.x = x;
        n.pad0 = 0;
        n.pad1 = 0.0;
        srow[row + 1] = n;
    }
    // D(0,j) scale: pre-scaled rows carry im of the first read row
    const double scale0 = (scaled && R > 0) ? 1.0 - p.eps[p.gcp[ro]] : 1.0;
    // LDS ops of one wave execute in order; only the compiler must not reorder across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const LdsView lds{srow};
    const bool group_head = (L == 32) && (lane == 32);

    const int nquads = (Nh + G - 1) / G;
    for (int quad = blockIdx.y; quad < nquads; quad += gridDim.y) {
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
        for (int w = 0; w < HapCols<K>::W; ++w) {
            hc.y[w] = 0u;
            hc.m[w] = 0u;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int col = l * K + k;
            const uint32_t y = (col < H) ? (uint32_t)p.hap_bases[ho + col] : 0u;
            const bool is_n = (y == 'N');
            lane_n |= is_n;
            hc.set(k, is_n ? 0u : y, is_n ? 0u : 0xffu);
        }
        const double c0 = p.initial_condition / (double)H * scale0;
        double s;
        if (scaled && __ballot(lane_n) == 0ull)
            s = p.exec_select ? sweep_fast<L, K, ROW_FAST_EXEC>(lds, R, l, group_head, hc, H, c0)
                              : sweep_fast<L, K, ROW_FAST>(lds, R, l, group_head, hc, H, c0);
        else  // rare: haplotype 'N' is a wildcard too (pair_hmm.rs:643), or a read with gcp == 0
            s = sweep_general<L, K>(lds, R, l, group_head, hc, H, c0, scaled);
#pragma unroll
        for (int off = L / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        if (l == 0 && hv) {
            const double v = log10(s) - p.initial_condition_log10;
            out_row[a] = v;
            if (!(v <= 0.0)) atomicOr(p.status, 1u);  // reference asserts result <= 0 (pair_hmm.rs:478-481)
        }
    }
}

#ifdef PHMM_WITH_GENERIC
// ---- generic any-shape fallback -----------------------------------------------------------------
// One thread per (read, haplotype) pair, two rolling rows of M/I/D in global scratch, interleaved by
// thread so that neighbouring threads touch neighbouring addresses.  Only used for shapes outside
// the register-resident kernel (haplotype > 64*KMAX columns or read too long for the LDS staging).
__global__ __launch_bounds__(256) void phmm_forward_generic(const GenericParams gp) {
    const ForwardParams &p = gp.f;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t W = (uint64_t)gp.max_h + 1;
    double *S = gp.scratch;
    auto at = [&](int arr, uint64_t j) -> double & { return S[((uint64_t)arr * W + j) * nthreads + tid]; };
    for (uint64_t pair = tid; pair < gp.n_pairs; pair += nthreads) {
        // item = last i with pair_first[i] <= pair
        uint32_t lo = 0, hi = p.n_items;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (gp.pair_first[mid] <= pair) lo = mid; else hi = mid;
        }
        const uint32_t r = p.class_reads ? p.class_reads[lo] : lo;
        const uint32_t a = (uint32_t)(pair - gp.pair_first[lo]);
        const uint32_t reg = p.read_region[r];
        const uint32_t ro = p.read_off[r];
        const int R = (int)(p.read_off[r + 1] - ro);
        const uint32_t h0 = p.region_hap_off[reg];
        const int Nh = (int)(p.region_hap_off[reg + 1] - h0);
        const uint32_t ho = p.hap_off[h0 + a];
        const int H = (int)(p.hap_off[h0 + a + 1] - ho);
        const double c = p.initial_condition / (double)H;
        int prv = 0, cur = 3;
        for (int j = 0; j <= H; ++j) {
            at(prv + 0, j) = 0.0;
            at(prv + 1, j) = 0.0;
            at(prv + 2, j) = c;
        }
        for (int i = 0; i < R; ++i) {
            const uint32_t x = p.read_bases[ro + i], q = p.base_q[ro + i], iq = p.ins_q[ro + i], dq = p.del_q[ro + i],
                           g = p.gcp[ro + i];
            const uint32_t mx = max(iq, dq), mn = min(iq, dq);
            const double mm = p.mm[((mx * (mx + 1)) >> 1) + mn], mi = p.eps[iq], md = p.eps[dq], ii = p.eps[g];
            const double im = 1.0 - ii, eq = p.eps[q], pm = 1.0 - eq;
            const double px = (x == 'N') ? pm : p.eps_mis[q];
            double dM = at(prv + 0, 0), dI = at(prv + 1, 0), dD = at(prv + 2, 0);
            double leftM = 0.0, leftD = 0.0;
            at(cur + 0, 0) = 0.0;
            at(cur + 1, 0) = 0.0;
            at(cur + 2, 0) = 0.0;
            for (int j = 1; j <= H; ++j) {
                const uint32_t y = p.hap_bases[ho + j - 1];
                const double uM = at(prv + 0, j), uI = at(prv + 1, j), uD = at(prv + 2, j);
                const double prior = (x == y || y == 'N') ? pm : px;
                double t = dM * mm;
                t = fma(dI, im, t);
                t = fma(dD, im, t);
                const double Mn = prior * t;
                const double In = fma(uI, ii, uM * mi);
                const double Dn = fma(leftD, ii, leftM * md);
                at(cur + 0, j) = Mn;
                at(cur + 1, j) = In;
                at(cur + 2, j) = Dn;
                dM = uM;
                dI = uI;
                dD = uD;
                leftM = Mn;
                leftD = Dn;
            }
            const int tmp = prv;
            prv = cur;
            cur = tmp;
        }
        double s = 0.0;
        for (int j = 1; j <= H; ++j) s += at(prv + 0, j) + at(prv + 1, j);
        const double v = log10(s) - p.initial_condition_log10;
        p.out[p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh + a] = v;
        if (!(v <= 0.0)) atomicOr(p.status, 1u);
    }
}

#endif  // PHMM_WITH_GENERIC

// ---- launch tables -------------------------------------------------------------------------------
#define PHMM_K_LIST(X, L)                                                                              \
    X(L, 2) X(L, 3) X(L, 4) X(L, 5) X(L, 6) X(L, 7) X(L, 8) X(L, 9) X(L, 10) X(L, 11) X(L, 12) X(L, 13) X(L, 14)  \
    X(L, 15) X(L, 16) X(L, 17) X(L, 18) X(L, 19) X(L, 20) X(L, 21) X(L, 22) X(L, 23) X(L, 24) X(L, 25) X(L, 26) \
    X(L, 27) X(L, 28) X(L, 29) X(L, 30) X(L, 31) X(L, 32)
// This file is compiled once per lanes-per-pair value (-DPHMM_L=16|32|64) so the three sets of 31
// instantiations build in parallel; the L=16 object also carries the generic kernel and the dispatcher.
#ifndef PHMM_L
#error "compile with -DPHMM_L=16|32|64"
#endif
#define PHMM_CAT2(a, b) a##b
#define PHMM_CAT(a, b) PHMM_CAT2(a, b)
hipError_t PHMM_CAT(launch_forward_L, PHMM_L)(int K, const ForwardParams &p, dim3 grid, int waves_per_block,
                                              size_t lds_bytes, hipStream_t stream) {
#define PHMM_CASE(LL, KK)                                                                          \
    if (K == KK) {                                                                                 \
        auto kern = phmm_forward<LL, KK>;                                                          \
        if (lds_bytes > 64 * 1024) {                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e != hipSuccess) return e;                                                         \
        }                                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(WAVE * waves_per_block), lds_bytes, stream, p);        \
        return hipGetLastError();                                                                  \
    }
    PHMM_K_LIST(PHMM_CASE, PHMM_L)
#undef PHMM_CASE
    return hipErrorInvalidValue;
}

#ifdef PHMM_WITH_GENERIC
hipError_t launch_forward_L16(int, const ForwardParams &, dim3, int, size_t, hipStream_t);
hipError_t launch_forward_L32(int, const ForwardParams &, dim3, int, size_t, hipStream_t);
hipError_t launch_forward_L64(int, const ForwardParams &, dim3, int, size_t, hipStream_t);

hipError_t launch_forward(int L, int K, const ForwardParams &p, dim3 grid, int waves_per_block, size_t lds_bytes,
                          hipStream_t stream) {
    if (L == 16) return launch_forward_L16(K, p, grid, waves_per_block, lds_bytes, stream);
    if (L == 32) return launch_forward_L32(K, p, grid, waves_per_block, lds_bytes, stream);
    if (L == 64) return launch_forward_L64(K, p, grid, waves_per_block, lds_bytes, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_generic(const GenericParams &gp, hipStream_t stream) {
    // scratch was sized for exactly this grid by the planner
    hipLaunchKernelGGL(phmm_forward_generic, dim3(gp.n_blocks), dim3(256), 0, stream, gp);
    return hipGetLastError();
}

const int kInstantiatedK[] = {2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                              18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32};
const int kNumInstantiatedK = sizeof(kInstantiatedK) / sizeof(int);
#endif  // PHMM_WITH_GENERIC

}  // namespace phmm
