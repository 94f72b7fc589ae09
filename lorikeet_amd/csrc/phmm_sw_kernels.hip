// gfx950 Smith-Waterman aligner: the reference's SmithWatermanAligner::align
// (src/smith_waterman/smith_waterman_aligner.rs:47-107: dispatch and exact-substring shortcut, :124-271 calculate_matrix,
// :273-443 calculate_cigar), all of it on the device, bit for bit -- the arithmetic is i32 and the tie rules are the
// reference's, so CIGAR and offset are EQUAL to the scalar arm's, not close to them.
//
// Mapping.  One wave works on one alignment at a time (a persistent grid of workers draws alignments from a counter).
// The matrix is (n+1) x (m+1), rows = reference, columns = alternate.  The 64 lanes own 64 consecutive COLUMNS (a
// strip); lane l walks down the rows one step behind lane l-1, so the wave sweeps anti-diagonals and every dependency
// of the recurrence is either in the lane's own registers or one lane to the left:
//   per column (registers):  sw[i-1][j], best_gap_v[j], gap_size_v[j]            (:140-141,196-211)
//   along the row (DPP wave_shr:1 from lane l-1):  sw[i][j-1], best_gap_h[i], gap_size_h[i]   (:142-143,220-233)
//   the diagonal sw[i-1][j-1] is last step's left value.
// Strips are processed left to right; what leaves a strip on its right edge -- three i32 per row -- waits in LDS for
// the next strip.  Nothing of the score matrix is ever stored except its last column and bottom row (the cells
// calculate_cigar starts from, :289-330); the backtrack matrix goes to HBM as int16 (0 = diagonal, +k = k rows up,
// -k = k columns left: the reference's own encoding, :257-266) in a skewed layout, slot [strip][step][lane], so that
// every wave step writes one contiguous 128-byte line.  Backtracking (one lane; it is a pointer chase) reads ~n+m of
// those entries and writes the CIGAR.
#include "phmm_internal.hpp"

namespace phmm {

namespace {

constexpr int32_t SW_LOW_INIT = INT32_MIN / 2;        // :137
constexpr int32_t SW_MATRIX_MIN_CUTOFF = -100000000;  // :31
enum : uint32_t { OP_M = 0, OP_I = 1, OP_D = 2, OP_S = 4 };
enum : int { ST_MATCH = 0, ST_INSERTION = 1, ST_DELETION = 2, ST_CLIP = 3 };

__device__ __forceinline__ int32_t shr1(int32_t v) {  // lane l <- lane l-1 (lane 0 gets 0)
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}

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

// One alignment by one wave.  Every early exit below is wave-uniform.
__device__ __forceinline__ void align_one(const SwParams &p, const uint32_t a, unsigned char *smem, int16_t *slab, const int lane) {
    // LDS: reference bytes | alternate bytes | strip edge (sw, best_gap_h, gap_size_h per row) | last column | bottom row
    uint8_t *s_ref = smem;
    uint8_t *s_alt = s_ref + p.lds_ref_bytes;
    int32_t *e_sw = reinterpret_cast<int32_t *>(s_alt + p.lds_alt_bytes);
    int32_t *e_bgh = e_sw + (p.max_ref + 1);
    int32_t *e_gsh = e_bgh + (p.max_ref + 1);
    int32_t *lastcol = e_gsh + (p.max_ref + 1);
    int32_t *bottom = lastcol + (p.max_ref + 1);
    const int32_t w_match = p.w_match, w_mismatch = p.w_mismatch, w_open = p.w_open, w_extend = p.w_extend;
    const bool edge_gaps = p.strategy == PHMM_SW_STRATEGY_INDEL || p.strategy == PHMM_SW_STRATEGY_LEADING_INDEL;  // :145
    {
        const uint32_t ro = p.ref_off[a], ao = p.alt_off[a];
        const int n = (int)(p.ref_off[a + 1] - ro), m = (int)(p.alt_off[a + 1] - ao);
        CigarOut cig{p.cigar + p.cigar_off[a], p.cigar_off[a + 1] - p.cigar_off[a]};
        if (n == 0 || m == 0) {  // the reference asserts (:65-68, :132-134)
            if (lane == 0) {
                p.n_cigar[a] = 0;
                p.alignment_offset[a] = 0;
                atomicOr(p.status, SW_STATUS_EMPTY);
            }
            return;
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < n; k += WAVE) s_ref[k] = p.ref_bases[ro + k];
        for (int k = lane; k < m; k += WAVE) s_alt[k] = p.alt_bases[ao + k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- exact substring: SoftClip / Ignore only (:72-81), the LAST occurrence (alignment_utils.rs:717-735) ----
        if (p.strategy == PHMM_SW_STRATEGY_SOFTCLIP || p.strategy == PHMM_SW_STRATEGY_IGNORE) {
            int found = -1;
            for (int r0 = n - m; r0 >= 0 && found < 0; r0 -= WAVE) {
                const int r = r0 - lane;
                bool ok = r >= 0;
                for (int q = 0; ok && q < m; ++q) ok = s_ref[r + q] == s_alt[q];
                const uint64_t hit = __ballot(ok);
                if (hit) found = r0 - (__ffsll((long long)hit) - 1);
            }
            if (found >= 0) {
                if (lane == 0) {
                    cig.push(make_element(ST_MATCH, (uint32_t)m));
                    cig.finish();
                    p.n_cigar[a] = cig.n;
                    p.alignment_offset[a] = found;
                    if (cig.n > cig.cap) atomicOr(p.status, SW_STATUS_CAPACITY);
                }
                return;
            }
        }

        // ---- calculate_matrix (:124-271), strip by strip -----------------------------------------------------------
        const int n_strips = (m + WAVE - 1) / WAVE;
        const int steps = n + WAVE - 1;
        for (int s = 0; s < n_strips; ++s) {
            const int j = s * WAVE + lane + 1;         // this lane's column
            const bool col_ok = j <= m;
            const int32_t b_base = col_ok ? (int32_t)s_alt[j - 1] : 0x1000;
            // row 0 of the matrix: gap penalties for InDel / LeadingInDel (:150-158), zeros otherwise
            auto row0 = [&](int jj) { return (edge_gaps && jj > 0) ? w_open + (jj - 1) * w_extend : 0; };
            int32_t up = row0(j);            // sw[i-1][j]
            int32_t diag = row0(j - 1);      // sw[i-1][j-1]
            int32_t bgv = SW_LOW_INIT, gsv = 0;
            int32_t o_sw = 0, o_bgh = 0, o_gsh = 0;  // what this lane hands to its right neighbour (row of the previous step)
            int16_t *bt = slab + (size_t)s * (size_t)(p.max_ref + WAVE) * WAVE + lane;
            for (int t = 0; t < steps; ++t) {
                const int i = t - lane + 1;  // this lane's row at this step
                int32_t l_sw = shr1(o_sw), l_bgh = shr1(o_bgh), l_gsh = shr1(o_gsh);
                const bool active = i >= 1 && i <= n;
                if (lane == 0 && active) {
                    if (s == 0) {            // column 0: gap penalties (:161-168) or zeros; no horizontal gap is open yet
                        l_sw = edge_gaps ? w_open + (i - 1) * w_extend : 0;
                        l_bgh = SW_LOW_INIT;
                        l_gsh = 0;
                    } else {                 // the right edge of the previous strip
                        l_sw = e_sw[i];
                        l_bgh = e_bgh[i];
                        l_gsh = e_gsh[i];
                    }
                }
                if (active) {
                    const int32_t a_base = (int32_t)s_ref[i - 1];
                    const int32_t step_diag = diag + (a_base == b_base ? w_match : w_mismatch);  // :194-199
                    int32_t prev_gap = up + w_open;                                              // :207
                    bgv += w_extend;
                    if (prev_gap > bgv) {
                        bgv = prev_gap;
                        gsv = 1;
                    } else {
                        gsv += 1;
                    }
                    const int32_t step_down = bgv, kd = gsv;
                    prev_gap = l_sw + w_open;                                                    // :229
                    int32_t bgh = l_bgh + w_extend, gsh;
                    if (prev_gap > bgh) {
                        bgh = prev_gap;
                        gsh = 1;
                    } else {
                        gsh = l_gsh + 1;
                    }
                    const int32_t step_right = bgh, ki = gsh;
                    int32_t cur, btr;
                    if (step_diag >= step_down && step_diag >= step_right) {                     // :250-266
                        cur = max(SW_MATRIX_MIN_CUTOFF, step_diag);
                        btr = 0;
                    } else if (step_right >= step_down) {
                        cur = max(SW_MATRIX_MIN_CUTOFF, step_right);
                        btr = -ki;
                    } else {
                        cur = max(SW_MATRIX_MIN_CUTOFF, step_down);
                        btr = kd;
                    }
                    bt[(size_t)t * WAVE] = (int16_t)btr;
                    diag = l_sw;
                    up = cur;
                    o_sw = cur;
                    o_bgh = bgh;
                    o_gsh = gsh;
                    if (lane == WAVE - 1) {  // leaves the strip: the next one picks it up at this row
                        e_sw[i] = cur;
                        e_bgh[i] = bgh;
                        e_gsh[i] = gsh;
                    }
                    if (j == m) lastcol[i] = cur;
                    if (i == n && col_ok) bottom[j] = cur;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __threadfence();  // the backtrack entries of all lanes are visible to lane 0

        // ---- calculate_cigar (:273-443): one lane ------------------------------------------------------------------
        if (lane == 0) {
            auto BT = [&](int i, int jj) -> int32_t {
                const int ss = (jj - 1) >> 6, ll = (jj - 1) & 63;
                return (int32_t)slab[(size_t)ss * (size_t)(p.max_ref + WAVE) * WAVE + (size_t)(i - 1 + ll) * WAVE + ll];
            };
            int p1 = 0, p2;
            int32_t max_score = INT32_MIN;
            int32_t segment_length = 0;
            if (p.strategy == PHMM_SW_STRATEGY_INDEL) {
                p1 = n;
                p2 = m;
            } else {
                p2 = m;
                for (int i = 1; i <= n; ++i) {        // rightmost column, `>=`: the lowest of equals (:303-309)
                    const int32_t cur = lastcol[i];
                    if (cur >= max_score) {
                        p1 = i;
                        max_score = cur;
                    }
                }
                if (p.strategy != PHMM_SW_STRATEGY_LEADING_INDEL) {
                    for (int jj = 1; jj <= m; ++jj) {  // bottom row (:316-330)
                        const int32_t cur = bottom[jj];
                        if (cur > max_score || (cur == max_score && abs(n - jj) < abs(p1 - p2))) {
                            p1 = n;
                            p2 = jj;
                            max_score = cur;
                            segment_length = m - jj;
                        }
                    }
                }
            }
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
            int32_t alignment_offset;
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
            cig.finish();
            p.n_cigar[a] = cig.n;
            p.alignment_offset[a] = alignment_offset;
            if (cig.n > cig.cap) atomicOr(p.status, SW_STATUS_CAPACITY);
        }
    }
}

// Workers (one wave each, a backtrack slab each) take alignments round robin.
__global__ __launch_bounds__(WAVE) void phmm_sw_align_kernel(const SwParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int16_t *slab = p.slab + (size_t)blockIdx.x * p.slab_stride;
    for (uint32_t a = blockIdx.x; a < p.n_alignments; a += gridDim.x) {
        align_one(p, a, smem, slab, (int)threadIdx.x);
        __builtin_amdgcn_s_barrier();  // (one wave per block: only a scheduling point between alignments)
    }
}

hipError_t launch_sw(const SwParams &p, uint32_t n_workers, size_t lds_bytes, hipStream_t stream) {
    if (!p.n_alignments) return hipSuccess;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(phmm_sw_align_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(phmm_sw_align_kernel, dim3(n_workers), dim3(WAVE), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace phmm
