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
// by the lane that owns it, the bottom row (:316-330) is kept in LDS.  The backtrack matrix goes to HBM as int16
// (0 = diagonal, +k = k rows up, -k = k columns left: the reference's own encoding, :257-266), two entries per dword,
// slot [strip][step][column pair][lane]: every store of a wave covers 256 contiguous bytes.  Backtracking is a pointer
// chase of ~n+m entries: one lane per alignment, four at a time per wave, writes the CIGAR.
#include "phmm_internal.hpp"

namespace phmm {

namespace {

constexpr int32_t SW_LOW_INIT = INT32_MIN / 2;        // :137
constexpr int32_t SW_MATRIX_MIN_CUTOFF = -100000000;  // :31
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
    uint32_t n = 0;
    __device__ void push(uint32_t e) {
        if (n < cap) slot[n] = e;
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

}  // namespace

constexpr int SW_L = 16;  // lanes per alignment

__device__ __forceinline__ int32_t row_shr1(int32_t v) {  // lane l <- lane l-1 inside each group of 16 (first lane: 0)
    return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
}

// Candidate start cells of the backtrack compare as the reference's scans do (:303-330): higher score first; among equal
// scores the smaller |p1 - p2|; among those the one met first (last column before bottom row, bottom row left to right).
struct Start {
    int32_t score, dist, order, p1, p2;
};
__device__ __forceinline__ bool better(const Start &a, const Start &b) {  // a beats b
    if (a.score != b.score) return a.score > b.score;
    if (a.dist != b.dist) return a.dist < b.dist;
    return a.order < b.order;
}

template <int K>
__global__ __launch_bounds__(WAVE) void phmm_sw_align_kernel(const SwParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
    // LDS of this group: reference | alternate | bottom row | (several strips only) strip edge: sw, best_gap_h, -gap_size_h
    const uint32_t gpb = p.groups_per_block;  // 4, or 1 when the sequences are so long that a block's LDS holds one alignment
    unsigned char *gbase = smem + (size_t)(g < (int)gpb ? g : 0) * p.lds_group_bytes;
    uint8_t *s_ref = gbase;
    uint8_t *s_alt = s_ref + p.lds_ref_bytes;
    int32_t *bottom = reinterpret_cast<int32_t *>(s_alt + p.lds_alt_bytes);
    int32_t *e_sw = bottom + (p.max_alt + 1);
    int32_t *e_bgh = e_sw + (p.max_ref + 1);
    int32_t *e_ngsh = e_bgh + (p.max_ref + 1);
    // backtrack storage of this block: dwords of two entries, laid out [strip][step][column pair][lane], so that every
    // store instruction of the wave writes 256 contiguous bytes
    uint32_t *slab = reinterpret_cast<uint32_t *>(p.slab) + (size_t)blockIdx.x * (p.slab_stride / 2);
    const int32_t w_match = p.w_match, w_mismatch = p.w_mismatch, w_open = p.w_open, w_extend = p.w_extend;
    const bool edge_gaps = p.strategy == PHMM_SW_STRATEGY_INDEL || p.strategy == PHMM_SW_STRATEGY_LEADING_INDEL;  // :145
    const int strip_cols = SW_L * K;
    const size_t strip_stride = (size_t)(p.max_ref + SW_L) * (K / 2) * WAVE;  // backtrack dwords of one strip
    auto row0 = [&](int jj) { return (edge_gaps && jj > 0) ? w_open + (jj - 1) * w_extend : 0; };  // :150-158

    for (uint32_t base = blockIdx.x * gpb; base < p.n_alignments; base += gridDim.x * gpb) {
        const uint32_t a = base + (uint32_t)g;
        const bool valid = (uint32_t)g < gpb && a < p.n_alignments;
        uint32_t ro = 0, ao = 0;
        int n = 0, m = 0;
        if (valid) {
            ro = p.ref_off[a];
            ao = p.alt_off[a];
            n = (int)(p.ref_off[a + 1] - ro);
            m = (int)(p.alt_off[a + 1] - ao);
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = l; k < n; k += SW_L) s_ref[k] = p.ref_bases[ro + k];
        for (int k = l; k < m; k += SW_L) s_alt[k] = p.alt_bases[ao + k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- exact substring: SoftClip / Ignore only (:72-81), the LAST occurrence (alignment_utils.rs:717-735) ----
        int found = -1;
        if (p.strategy == PHMM_SW_STRATEGY_SOFTCLIP || p.strategy == PHMM_SW_STRATEGY_IGNORE) {
            int r0 = valid ? n - m : -1;
            while (__any(found < 0 && r0 >= 0)) {
                const int r = r0 - l;
                bool ok = found < 0 && r0 >= 0 && r >= 0;
                for (int q = 0; ok && q < m; ++q) ok = s_ref[r + q] == s_alt[q];
                const uint32_t hit = (uint32_t)(__ballot(ok) >> (lane & 48)) & 0xffffu;
                if (hit && found < 0) found = r0 - (__ffs((int)hit) - 1);
                r0 -= SW_L;
            }
        }
        const bool dp = valid && found < 0;  // this group runs the matrix

        // ---- calculate_matrix (:124-271) -----------------------------------------------------------------------------
        const int my_strips = dp ? (m + strip_cols - 1) / strip_cols : 0;
        int n_strips = my_strips, n_max = dp ? n : 0;
#pragma unroll
        for (int o = 32; o >= 16; o >>= 1) {  // over the four groups
            n_strips = max(n_strips, __shfl_xor(n_strips, o, WAVE));
            n_max = max(n_max, __shfl_xor(n_max, o, WAVE));
        }
        // the last column's best cell, tracked by the lane that owns column m (`>=`: the lowest of equals, :303-309)
        const int lm = ((m - 1) % strip_cols) / K, km = (m - 1) % K, sm = (m - 1) / strip_cols;
        int32_t lc_score = INT32_MIN, lc_row = 0;
        for (int s = 0; s < n_strips; ++s) {
            const bool strip_on = dp && s < my_strips;
            const int j0 = s * strip_cols + l * K;  // columns j0+1 .. j0+K
            int32_t up[K], bgv[K], gsv[K], bb[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = j0 + k + 1;
                bb[k] = (strip_on && j <= m) ? (int32_t)s_alt[j - 1] : 0x1000;
                up[k] = row0(j);
                bgv[k] = SW_LOW_INIT;
                gsv[k] = 0;
            }
            int32_t diag = row0(j0);                     // sw[i-1][j0]
            int32_t o_sw = 0, o_bgh = 0, o_ngsh = 0;     // what this lane hands to its right neighbour (row of the previous step)
            uint32_t *bt = slab + (size_t)s * strip_stride + lane;
            const int steps = n_max + SW_L - 1;
            for (int t = 0; t < steps; ++t) {
                const int i = t - l + 1;                 // this lane's row at this step
                int32_t left = row_shr1(o_sw), h_bg = row_shr1(o_bgh), h_ngs = row_shr1(o_ngsh);
                const bool active = strip_on && i >= 1 && i <= n;
                if (active) {
                    if (l == 0) {
                        if (s == 0) {                    // column 0: gap penalties (:161-168) or zeros; no horizontal gap yet
                            left = edge_gaps ? w_open + (i - 1) * w_extend : 0;
                            h_bg = SW_LOW_INIT;
                            h_ngs = 0;
                        } else {                         // the right edge of the previous strip
                            left = e_sw[i];
                            h_bg = e_bgh[i];
                            h_ngs = e_ngsh[i];
                        }
                    }
                    const int32_t a_base = (int32_t)s_ref[i - 1];
                    const int32_t diag_next = left;      // sw[i][j0]: the diagonal of this lane's first column, next row
                    int32_t d = diag;
                    int32_t btr[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int32_t step_diag = d + (a_base == bb[k] ? w_match : w_mismatch);   // :194-199
                        const int32_t pv = up[k] + w_open;                                         // :207-218
                        const int32_t ev = bgv[k] + w_extend;
                        gsv[k] = pv > ev ? 1 : gsv[k] + 1;
                        bgv[k] = max(pv, ev);
                        const int32_t ph = left + w_open;                                          // :229-240
                        const int32_t eh = h_bg + w_extend;
                        h_ngs = ph > eh ? -1 : h_ngs - 1;                                          // minus the gap length
                        h_bg = max(ph, eh);
                        const int32_t gap = max(h_bg, bgv[k]);
                        // priority: diagonal, then right (horizontal), then down (:250-266)
                        btr[k] = step_diag >= gap ? 0 : (h_bg >= bgv[k] ? h_ngs : gsv[k]);
                        const int32_t cur = max(SW_MATRIX_MIN_CUTOFF, max(step_diag, gap));
                        d = up[k];
                        up[k] = cur;
                        left = cur;
                    }
                    uint32_t *row_bt = bt + (size_t)t * (K / 2) * WAVE;
#pragma unroll
                    for (int k = 0; k + 1 < K; k += 2)
                        row_bt[(k / 2) * WAVE] = (uint32_t)(uint16_t)btr[k] | ((uint32_t)(uint16_t)btr[k + 1] << 16);
                    diag = diag_next;
                    o_sw = left;
                    o_bgh = h_bg;
                    o_ngsh = h_ngs;
                    if (l == SW_L - 1 && s + 1 < my_strips) {  // leaves the strip: the next one picks it up at this row
                        e_sw[i] = left;
                        e_bgh[i] = h_bg;
                        e_ngsh[i] = h_ngs;
                    }
                    if (s == sm && l == lm) {
                        int32_t v = up[0];
#pragma unroll
                        for (int k = 1; k < K; ++k) v = (k == km) ? up[k] : v;
                        if (v >= lc_score) {
                            lc_score = v;
                            lc_row = i;
                        }
                    }
                    if (i == n) {
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            if (j0 + k + 1 <= m) bottom[j0 + k + 1] = up[k];
                    }
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
                const int src = (lane & 48) | lm;
                const int32_t sc = __shfl(lc_score, src, WAVE), rw = __shfl(lc_row, src, WAVE);
                best = Start{sc, abs(rw - m), 0, rw, m};
                if (p.strategy != PHMM_SW_STRATEGY_LEADING_INDEL) {
                    for (int j = l + 1; j <= m; j += SW_L) {  // bottom row, every lane a share of the columns
                        const Start c{bottom[j], abs(n - j), j, n, j};
                        if (better(c, best)) best = c;
                    }
                }
            }
        }
        if (p.strategy != PHMM_SW_STRATEGY_INDEL && p.strategy != PHMM_SW_STRATEGY_LEADING_INDEL) {
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {  // best of the group
                Start c;
                c.score = __shfl_xor(best.score, o, WAVE);
                c.dist = __shfl_xor(best.dist, o, WAVE);
                c.order = __shfl_xor(best.order, o, WAVE);
                c.p1 = __shfl_xor(best.p1, o, WAVE);
                c.p2 = __shfl_xor(best.p2, o, WAVE);
                if (better(c, best)) best = c;
            }
        }
        if (dp && best.order > 0) segment_length = m - best.p2;  // a bottom-row cell: the end of the alternate overhangs (:327)
        __threadfence();  // every lane's backtrack entries are visible to the lane that walks them

        // ---- calculate_cigar (:332-443): one lane per alignment ------------------------------------------------------
        if (valid && l == 0) {
            CigarOut cig{p.cigar + p.cigar_off[a], p.cigar_off[a + 1] - p.cigar_off[a]};
            int32_t alignment_offset = 0;
            if (n == 0 || m == 0) {  // the reference asserts (:65-68, :132-134); the host refuses such input beforehand
                atomicOr(p.status, SW_STATUS_EMPTY);
            } else if (found >= 0) {
                cig.push(make_element(ST_MATCH, (uint32_t)m));
                alignment_offset = found;
            } else {
                auto BT = [&](int i, int jj) -> int32_t {
                    const int ss = (jj - 1) / strip_cols, cc = (jj - 1) % strip_cols, ll = cc / K, kk = cc % K;
                    const uint32_t w = slab[(size_t)ss * strip_stride + ((size_t)(i - 1 + ll) * (K / 2) + kk / 2) * WAVE + (lane & 48) + ll];
                    return (int32_t)(int16_t)((kk & 1) ? (w >> 16) : (w & 0xffffu));
                };
                int p1 = best.p1, p2 = best.p2;
                if (segment_length > 0 && p.strategy == PHMM_SW_STRATEGY_SOFTCLIP) {
                    cig.push(make_element(ST_CLIP, (uint32_t)segment_length));
                    segment_length = 0;
                }
                int state = ST_MATCH;
                for (;;) {
                    const int32_t btr = BT(p1, p2);
                    int new_state;
                    int32_t step_length = 1;
                    if (btr > 0) {
                        new_state = ST_DELETION;
                        step_length = btr;
                    } else if (btr < 0) {
                        new_state = ST_INSERTION;
                        step_length = -btr;
                    } else {
                        new_state = ST_MATCH;
                    }
                    if (new_state == ST_MATCH) {
                        p1 -= 1;
                        p2 -= 1;
                    } else if (new_state == ST_INSERTION) {
                        p2 -= step_length;
                    } else {
                        p1 -= step_length;
                    }
                    if (new_state == state) {
                        segment_length += step_length;
                    } else {
                        if (segment_length > 0) cig.push(make_element(state, (uint32_t)segment_length));
                        segment_length = step_length;
                        state = new_state;
                    }
                    if (p1 <= 0 || p2 <= 0) break;
                }
                if (p.strategy == PHMM_SW_STRATEGY_SOFTCLIP) {
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
            cig.finish();
            p.n_cigar[a] = cig.n;
            p.alignment_offset[a] = alignment_offset;
            if (cig.n > cig.cap) atomicOr(p.status, SW_STATUS_CAPACITY);
        }
        __builtin_amdgcn_s_barrier();  // (one wave per block: a scheduling point between rounds)
    }
}

#define PHMM_SW_K_LIST(X) X(2) X(4) X(6) X(8) X(10) X(12) X(16) X(20) X(24) X(32)
const int kSwK[] = {2, 4, 6, 8, 10, 12, 16, 20, 24, 32};
const int kNumSwK = sizeof(kSwK) / sizeof(int);

hipError_t launch_sw(int K, const SwParams &p, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream) {
    if (!p.n_alignments) return hipSuccess;
#define PHMM_CASE(KK)                                                                                              \
    if (K == KK) {                                                                                                 \
        auto kern = phmm_sw_align_kernel<KK>;                                                                      \
        if (lds_bytes > 64 * 1024) {                                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);        \
            if (e != hipSuccess) return e;                                                                         \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(WAVE), lds_bytes, stream, p);                                \
        return hipGetLastError();                                                                                  \
    }
    PHMM_SW_K_LIST(PHMM_CASE)
#undef PHMM_CASE
    return hipErrorInvalidValue;
}

}  // namespace phmm
