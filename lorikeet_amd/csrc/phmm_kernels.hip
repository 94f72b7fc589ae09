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
#include "phmm_device.hpp"


namespace phmm {

template <int L, int K>
__global__ __launch_bounds__(WAVE *MAX_WAVES_PER_BLOCK, (K <= PHMM_TWO_WAVE_MAX_K ? 2 : 1)) void phmm_forward(const ForwardParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t item = blockIdx.x * (blockDim.x >> 6) + wave;
    if (item >= p.n_items) return;  // wave-uniform
    const uint32_t r = p.class_reads ? p.class_reads[item] : item;
    if (p.redo && p.redo[r] == 0) return;  // f32-first mode: this launch only redoes the flagged reads (wave-uniform)
    forward_read<L, K>(p, r, (int)blockIdx.y, (int)gridDim.y, p.cnd_select != 0, smem + (size_t)wave * p.lds_rows * sizeof(RowConst));
}

// ---- launch tables -------------------------------------------------------------------------------
#define PHMM_K_LIST(X, L)                                                                              \
    X(L, 2) X(L, 3) X(L, 4) X(L, 5) X(L, 6) X(L, 7) X(L, 8) X(L, 9) X(L, 10) X(L, 11) X(L, 12) X(L, 13) X(L, 14)  \
    X(L, 15) X(L, 16) X(L, 17) X(L, 18) X(L, 19) X(L, 20) X(L, 21) X(L, 22) X(L, 23) X(L, 24) X(L, 25) X(L, 26) \
    X(L, 27) X(L, 28) X(L, 29) X(L, 30) X(L, 31) X(L, 32)
// This file is compiled once per lanes-per-pair value (-DPHMM_L=16|32|64) so the three sets of 31
// instantiations build in parallel; the L=16 object also carries the dispatcher
// (the generic any-shape kernel lives in phmm_exact_kernels.hip).
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

const int kInstantiatedK[] = {2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                              18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32};
const int kNumInstantiatedK = sizeof(kInstantiatedK) / sizeof(int);
#endif  // PHMM_WITH_GENERIC

}  // namespace phmm
