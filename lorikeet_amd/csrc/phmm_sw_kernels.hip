// gfx950 Smith-Waterman aligner: the reference's SmithWatermanAligner::align
// (src/smith_waterman/smith_waterman_aligner.rs:47-107: dispatch and exact-substring shortcut, :124-271 calculate_matrix,
// :273-443 calculate_cigar), all of it on the device, bit for bit -- the arithmetic is i32 and the tie rules are the
// reference's, so CIGAR and offset are EQUAL to the scalar arm's, not close to them.
//
// Mapping (the PairHMM kernels' shape).  The matrix is (n+1) x (m+1), rows = reference, columns = alternate.  A group
// of 16 lanes owns one alignment, four alignments share a wave; lane l of a group keeps K consecutive COLUMNS in
// registers and walks down the rows one step behind lane l-1, so a group sweeps anti-diagonals of K-column blocks
// and every dependency of the recurrence is in the lane's own registers or one lane to the left:
//   per column (registers):  sw[i-1][j], best_gap_v[j], gap_size_v[j]                                (:140-141,196-211)
//   along the row: sw[i][j-1], best_gap_h[i], gap_size_h[i] run through the K cells of a lane and on to the next lane
//                  (DPP row_shr:1)                                                                   (:142-143,220-233)
//   the diagonal sw[i-1][j-1] is the previous column's old value (for a lane's first column: last step's left value).
// 16 x K columns make a strip (K is chosen so that one strip covers the batch's longest alternate sequence, up to 512
// columns; beyond that strips follow each other and what leaves one on its right edge -- three i32 per row -- waits
// in LDS for the next).  Nothing of the score matrix is stored: the best cell of the last column (:303-309) is tracked
// by the lane that owns it, the bottom row (:316-330) is kept in LDS.
// Backtrack: the reference stores 0 / +k / -k per cell (diagonal, k rows up, k columns left, :257-266), k being the
// length of the best gap ending there (gap_size_v / gap_size_h, :207-240).  Here a cell leaves FOUR BITS: which of the
// three candidates won (2 bits) and, for each direction, whether its best gap OPENS at this cell (1 = the `prev_gap >
// best_gap` branch).  The gap length is recovered while backtracking -- k(i,j) = 1 if the gap opens at (i,j), else
// 1 + k of the previous cell of that column / row -- so the kernel keeps no gap sizes at all, and the matrix in HBM is
// 2 dwords per lane and step (3 - 4 when K > 16) laid out [strip][step][lane][dword]: one vector store per lane and step,
// a wave's store covers 64 x NW x 4 contiguous bytes.  (The stores cost nothing: tools/ubench/store_cost.hip.)
// Scores are carried times four, the low two bits naming the candidate (diagonal 2 > right 1 > down 0): one v_max3
// both picks the value and resolves ties in the reference's priority order (:250-266), and v_alignbit shifts the two
// bits into the lane's flag word.  (The host side bounds the parameters so that nothing overflows and the reference's
// MATRIX_MIN_CUTOFF clamp, :31, can never be active; beyond that bound the WIDE instance carries scores as they are.)
// Backtracking: the lanes of an alignment fetch 32 cells down the diagonal per round trip (device-scope loads: the flags
// are in L2 / HBM, every fetch is a dependent read) and take the run of diagonal steps among them in one go; lane 0
// writes the CIGAR.
// Instances: <lanes per alignment, columns per lane>, TR (sweep along the alternate: small calls), WIDE (un-scaled scores with
// the reference's clamp), EXT (rows in device memory: sequences beyond LDS), LITE (candidate tags only: the first of two
// passes where gaps are rare) -- see the template below.
#include "phmm_sw_device.hpp"

namespace phmm {

using namespace swdev;

template <int SW_L, int K, bool TR = false, bool WIDE = false, bool EXT = false, bool LITE = false>
__global__ __launch_bounds__(WAVE) PHMM_SW_OCCUPANCY(K)
void phmm_sw_align_kernel(const SwParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    sw_align_body<SW_L, K, TR, WIDE, EXT, LITE>(p, smem, blockIdx.x, gridDim.x, blockIdx.x);
}

// The alignments a second pass has redone, gathered for the host: entry t = { alignment, n_cigar, offset, slot[cap] } for the
// first `max_entries` of the list (a call in pieces fetches every piece's results as soon as its first pass is done; what the
// one second pass at the end of the call changes comes back this way, in one small copy).
__global__ __launch_bounds__(WAVE) void phmm_sw_gather_kernel(const uint32_t *todo, const uint32_t *todo_count, const uint32_t *n_cigar,
                                                             const int32_t *alignment_offset, const uint32_t *cigar, const uint64_t *cigar_off,
                                                             uint32_t cap, uint32_t max_entries, uint32_t *out) {
    const uint32_t n = min(*todo_count, max_entries);
    for (uint32_t t = blockIdx.x; t < n; t += gridDim.x) {
        const uint32_t a = todo[t];
        uint32_t *e = out + (size_t)t * (3 + cap);
        if (threadIdx.x == 0) {
            e[0] = a;
            e[1] = n_cigar[a];
            e[2] = (uint32_t)alignment_offset[a];
        }
        const uint64_t c0 = cigar_off[a], slot = cigar_off[a + 1] - c0;
        for (uint32_t i = threadIdx.x; i < cap; i += WAVE) e[3 + i] = i < slot ? cigar[c0 + i] : 0u;
    }
}

hipError_t launch_sw_gather(const uint32_t *todo, const uint32_t *todo_count, const uint32_t *n_cigar, const int32_t *alignment_offset,
                            const uint32_t *cigar, const uint64_t *cigar_off, uint32_t cap, uint32_t max_entries, uint32_t *out, hipStream_t stream) {
    hipLaunchKernelGGL(phmm_sw_gather_kernel, dim3(std::min<uint32_t>(max_entries, 1024)), dim3(WAVE), 0, stream, todo, todo_count, n_cigar, alignment_offset,
                       cigar, cigar_off, cap, max_entries, out);
    return hipGetLastError();
}

// instantiated <lanes per alignment, columns per lane>; the host side picks the pair (phmm_sw.cpp)
#define PHMM_SW_LIST(X)                                                                                                \
    X(16, 2) X(16, 4) X(16, 6) X(16, 8) X(16, 10) X(16, 12) X(16, 14) X(16, 16) X(16, 20) X(16, 24) X(16, 28) X(16, 32) \
    X(8, 4) X(8, 8) X(8, 12) X(8, 16) X(8, 19) X(8, 22) X(8, 26) X(8, 32)                                             \
    X(32, 3) X(32, 4) X(32, 5) X(32, 6) X(32, 8) X(32, 12) X(32, 16) X(64, 2) X(64, 3) X(64, 4) X(64, 6) X(64, 8)
const int kSwK16[] = {2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32};
const int kNumSwK16 = sizeof(kSwK16) / sizeof(int);
const int kSwK8[] = {4, 8, 12, 16, 19, 22, 26, 32};
const int kNumSwK8 = sizeof(kSwK8) / sizeof(int);
const int kSwK32[] = {3, 4, 5, 6, 8, 12, 16};
const int kNumSwK32 = sizeof(kSwK32) / sizeof(int);
const int kSwK64[] = {2, 3, 4, 6, 8};
const int kNumSwK64 = sizeof(kSwK64) / sizeof(int);
// ... and <64 lanes, rows per lane> of the sweep along the alternate sequence (small calls, reference longer than alternate)
#define PHMM_SW_LIST_T(X) X(64, 2) X(64, 3) X(64, 4) X(64, 5) X(64, 6) X(64, 8)
const int kSwK64T[] = {2, 3, 4, 5, 6, 8};
const int kNumSwK64T = sizeof(kSwK64T) / sizeof(int);

// blocks (of one wave) of this instance a CU holds at once, by registers and LDS
template <typename Kern>
static int blocks_per_cu_of(Kern kern, size_t lds_bytes) {
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return 0;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, WAVE, lds_bytes) != hipSuccess) nb = 0;
    return nb;
}
template <typename Kern>
static hipError_t launch_of(Kern kern, const SwParams &p, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(WAVE), lds_bytes, stream, p);
    return hipGetLastError();
}

// the special instances (sw_variant): one geometry each -- 16 lanes x 16 columns wide, 16 x 32 with rows in device memory
int sw_blocks_per_cu(int L, int K, size_t lds_bytes, bool transposed, int variant) {
    if (variant == SW_WIDE) return blocks_per_cu_of(phmm_sw_align_kernel<16, 16, false, true, false>, lds_bytes);
    if (variant == SW_EXT) return blocks_per_cu_of(phmm_sw_align_kernel<16, 32, false, false, true>, lds_bytes);
    if (variant == (SW_WIDE | SW_EXT)) return blocks_per_cu_of(phmm_sw_align_kernel<16, 16, false, true, true>, lds_bytes);
#define PHMM_CASE_T(LL, KK, TT)                                                                                              \
    if (L == LL && K == KK && transposed == TT)                                                                              \
        return variant == SW_LITE ? blocks_per_cu_of(phmm_sw_align_kernel<LL, KK, TT, false, false, true>, lds_bytes)       \
                                  : blocks_per_cu_of(phmm_sw_align_kernel<LL, KK, TT>, lds_bytes);
#define PHMM_CASE(LL, KK) PHMM_CASE_T(LL, KK, false)
#define PHMM_CASE_TR(LL, KK) PHMM_CASE_T(LL, KK, true)
    PHMM_SW_LIST(PHMM_CASE)
    PHMM_SW_LIST_T(PHMM_CASE_TR)
#undef PHMM_CASE
#undef PHMM_CASE_TR
#undef PHMM_CASE_T
    return 0;
}

hipError_t launch_sw(int L, int K, bool transposed, int variant, const SwParams &p, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream) {
    if (!p.todo && p.n_alignments <= p.a_begin) return hipSuccess;
    if (variant == SW_LITE && (!p.todo_out || !p.todo_out_count || p.todo)) return hipErrorInvalidValue;
    if (variant != SW_PLAIN && variant != SW_LITE) {
        if (L != 16 || K != ((variant & SW_WIDE) ? 16 : 32) || transposed || ((variant & SW_EXT) != 0) != (p.ext != nullptr)) return hipErrorInvalidValue;
        if (variant == SW_WIDE) return launch_of(phmm_sw_align_kernel<16, 16, false, true, false>, p, n_blocks, lds_bytes, stream);
        if (variant == SW_EXT) return launch_of(phmm_sw_align_kernel<16, 32, false, false, true>, p, n_blocks, lds_bytes, stream);
        return launch_of(phmm_sw_align_kernel<16, 16, false, true, true>, p, n_blocks, lds_bytes, stream);
    }
    if (p.ext) return hipErrorInvalidValue;
#define PHMM_CASE_T(LL, KK, TT)                                                                                                  \
    if (L == LL && K == KK && transposed == TT)                                                                                  \
        return variant == SW_LITE ? launch_of(phmm_sw_align_kernel<LL, KK, TT, false, false, true>, p, n_blocks, lds_bytes, stream) \
                                  : launch_of(phmm_sw_align_kernel<LL, KK, TT>, p, n_blocks, lds_bytes, stream);
#define PHMM_CASE(LL, KK) PHMM_CASE_T(LL, KK, false)
#define PHMM_CASE_TR(LL, KK) PHMM_CASE_T(LL, KK, true)
    PHMM_SW_LIST(PHMM_CASE)
    PHMM_SW_LIST_T(PHMM_CASE_TR)
#undef PHMM_CASE
#undef PHMM_CASE_TR
#undef PHMM_CASE_T
    return hipErrorInvalidValue;
}

}  // namespace phmm
