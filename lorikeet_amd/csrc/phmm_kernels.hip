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

struct RowConst {  // per read row, staged in LDS
    double mm, mi, md, ii, eq, px;
    uint32_t x;
};

struct LdsView {
    const double *mm, *mi, *md, *ii, *eq, *px;
    const uint8_t *x;
    __device__ __forceinline__ RowConst load(int row) const {
        RowConst c;
        c.mm = mm[row];
        c.mi = mi[row];
        c.md = md[row];
        c.ii = ii[row];
        c.eq = eq[row];
        c.px = px[row];
        c.x = x[row];
        return c;
    }
};

// One (read x up-to-64/L haplotypes) sweep.  Returns this lane's partial of sum_j M[R][j]+I[R][j].
template <int L, int K, bool HAPN>
__device__ __forceinline__ double sweep(const LdsView &lds, const int R, const int l, const bool group_head,
                                        const uint32_t (&yc)[K], const int H, const double c) {
    double Mp[K], Ip[K], Dp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Mp[k] = 0.0;
        Ip[k] = 0.0;
        Dp[k] = c;  // D[0][j] = 2^1020 / H for every column (pair_hmm.rs:515-529)
    }
    // (row i-1) values of the left neighbour's last column; row 0 there is (0, 0, c)
    double plM = 0.0, plI = 0.0, plD = c;

    const int nsteps = R + L - 1;
    int row = -l;  // 0-based read row this lane works on at step t (= t - l)
    const int rmax = R - 1;
    RowConst cur = lds.load(max(min(row, rmax), 0));
    for (int t = 0; t < nsteps; ++t) {
        const int nrow = row + 1;
        const RowConst nxt = lds.load(max(min(nrow, rmax), 0));  // one step ahead
        // left neighbour's last column after ITS previous step == row `row` there
        const double lM = from_left<L>(Mp[K - 1], group_head);
        const double lI = from_left<L>(Ip[K - 1], group_head);
        const double lD = from_left<L>(Dp[K - 1], group_head);
        if (row >= 0 && row < R) {
            const double im = 1.0 - cur.ii;  // qual_to_prob(gcp)
            const double pm = 1.0 - cur.eq;  // qual_to_prob(q)
            // Pass 1, columns right-to-left so every register is updated in place: I(i,k) reads the old
            // M/I of column k, then M(i,k) overwrites M[k] using the still-old column k-1.
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                bool match;
                if constexpr (HAPN)
                    match = (cur.x & (yc[k] >> 8)) == (yc[k] & 0xffu);
                else
                    match = cur.x == yc[k];
                const double prior = match ? pm : cur.px;
                Ip[k] = fma(Ip[k], cur.ii, Mp[k] * cur.mi);  // I(i,k) = M(i-1,k)*mi + I(i-1,k)*ii
                const double dM = k ? Mp[k - 1] : plM;       // (i-1, k-1)
                const double dI = k ? Ip[k - 1] : plI;
                const double dD = k ? Dp[k - 1] : plD;
                double a = dM * cur.mm;
                a = fma(dI, im, a);
                a = fma(dD, im, a);
                Mp[k] = prior * a;
            }
            // Pass 2, left-to-right: the serial D chain D(i,k) = M(i,k-1)*md + D(i,k-1)*dd.
            double leftM = lM, leftD = lD;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                Dp[k] = fma(leftD, cur.ii, leftM * cur.md);
                leftM = Mp[k];
                leftD = Dp[k];
            }
        }
        plM = lM;
        plI = lI;
        plD = lD;
        row = nrow;
        cur = nxt;
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (l * K + k < H) s += Mp[k] + Ip[k];
    return s;
}

template <int L, int K>
__global__ __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK) void phmm_forward(const ForwardParams p) {
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

    // ---- stage this read's per-row constants in wave-private LDS (SoA, conflict-free) ----------
    const uint32_t rows = p.lds_rows;  // multiple of 8
    unsigned char *base = smem + (size_t)wave * rows * 56u;
    double *s_mm = reinterpret_cast<double *>(base);
    double *s_mi = s_mm + rows, *s_md = s_mi + rows, *s_ii = s_md + rows, *s_eq = s_ii + rows, *s_px = s_eq + rows;
    uint8_t *s_x = reinterpret_cast<uint8_t *>(s_px + rows);
    for (int row = lane; row < R; row += WAVE) {
        const uint32_t x = p.read_bases[ro + row];
        const uint32_t q = p.base_q[ro + row];
        const uint32_t iq = p.ins_q[ro + row];
        const uint32_t dq = p.del_q[ro + row];
        const uint32_t g = p.gcp[ro + row];
        const uint32_t mx = max(iq, dq), mn = min(iq, dq);
        const double eq = p.eps[q];
        s_mm[row] = p.mm[((mx * (mx + 1)) >> 1) + mn];  // pair_hmm_model.rs:442-461
        s_mi[row] = p.eps[iq];
        s_md[row] = p.eps[dq];
        s_ii[row] = p.eps[g];
        s_eq[row] = eq;
        s_px[row] = (x == 'N') ? (1.0 - eq) : p.eps_mis[q];  // read 'N' matches everything (pair_hmm.rs:643)
        s_x[row] = (uint8_t)x;
    }
    // LDS ops of one wave execute in order; only the compiler must not reorder across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const LdsView lds{s_mm, s_mi, s_md, s_ii, s_eq, s_px, s_x};
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
        uint32_t yc[K];
        bool lane_n = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int col = l * K + k;
            const uint32_t y = (col < H) ? (uint32_t)p.hap_bases[ho + col] : 0u;
            yc[k] = y;
            lane_n |= (y == 'N');
        }
        const double c = p.initial_condition / (double)H;
        double s;
        if (__ballot(lane_n) != 0ull) {  // rare: haplotype 'N' is a wildcard too
#pragma unroll
            for (int k = 0; k < K; ++k) yc[k] = (yc[k] == 'N') ? 0u : (yc[k] | 0xff00u);
            s = sweep<L, K, true>(lds, R, l, group_head, yc, H, c);
        } else {
            s = sweep<L, K, false>(lds, R, l, group_head, yc, H, c);
        }
#pragma unroll
        for (int off = L / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        if (l == 0 && hv) {
            const double v = log10(s) - p.initial_condition_log10;
            out_row[a] = v;
            if (!(v <= 0.0)) atomicOr(p.status, 1u);  // reference asserts result <= 0 (pair_hmm.rs:478-481)
        }
    }
}

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

// ---- launch tables -------------------------------------------------------------------------------
#define PHMM_K_LIST(X, L) X(L, 2) X(L, 4) X(L, 6) X(L, 8) X(L, 10) X(L, 13) X(L, 16) X(L, 19) X(L, 22) X(L, 25) X(L, 28) X(L, 32)
const int kInstantiatedK[] = {2, 4, 6, 8, 10, 13, 16, 19, 22, 25, 28, 32};
const int kNumInstantiatedK = sizeof(kInstantiatedK) / sizeof(int);

hipError_t launch_forward(int L, int K, const ForwardParams &p, dim3 grid, int waves_per_block, size_t lds_bytes,
                          hipStream_t stream) {
#define PHMM_CASE(LL, KK)                                                                          \
    if (L == LL && K == KK) {                                                                      \
        auto kern = phmm_forward<LL, KK>;                                                          \
        if (lds_bytes > 64 * 1024) {                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e != hipSuccess) return e;                                                         \
        }                                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(WAVE * waves_per_block), lds_bytes, stream, p);        \
        return hipGetLastError();                                                                  \
    }
    PHMM_K_LIST(PHMM_CASE, 16)
    PHMM_K_LIST(PHMM_CASE, 32)
    PHMM_K_LIST(PHMM_CASE, 64)
#undef PHMM_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_generic(const GenericParams &gp, hipStream_t stream) {
    // scratch was sized for exactly this grid by the planner
    hipLaunchKernelGGL(phmm_forward_generic, dim3(gp.n_blocks), dim3(256), 0, stream, gp);
    return hipGetLastError();
}

}  // namespace phmm
