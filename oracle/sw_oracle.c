/*
 * sw_oracle.c -- CPU ORACLE for the Smith-Waterman aligner.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as pairhmm_oracle.c).
 *
 * Plain-C restatement of the reference's SCALAR arm, statement by statement:
 *   src/smith_waterman/smith_waterman_aligner.rs:47-107   SmithWatermanAligner::align (AVXMode::None arm), including the
 *                                                        exact-substring shortcut for SoftClip / Ignore
 *   :124-271   calculate_matrix   (i32 DP with the linear-gap "best gap so far" optimisation, backtrack matrix)
 *   :273-443   calculate_cigar    (start cell selection per OverhangStrategy, backtrack, overhang handling)
 *   src/reads/alignment_utils.rs:717-735   last_index_of
 * The reference's vector arm is gkl::smithwaterman::align (not vendored); the reference's own test
 * tests/smith_waterman_aligner_unit_tests.rs:999-1103 asserts it equal to this scalar arm.
 * Pinned by the asserted cases of that test file (tests/golden/smith_waterman_cases.json, tests/test_sw_oracle.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* gkl::smithwaterman::OverhangStrategy, in the order the C ABI numbers them (include/phmm.h) */
enum { SW_SOFTCLIP = 0, SW_INDEL = 1, SW_LEADING_INDEL = 2, SW_IGNORE = 3 };
/* BAM CIGAR op codes */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_S = 4 };
enum { ST_MATCH, ST_INSERTION, ST_DELETION, ST_CLIP };

#define MATRIX_MIN_CUTOFF (-100000000) /* :31 */

static uint32_t make_element(int state, uint32_t length) { /* :445-452 */
    static const uint32_t op[4] = {OP_M, OP_I, OP_D, OP_S};
    return (length << 4) | op[state];
}

/* alignment_utils.rs:717-735 */
static long last_index_of(const uint8_t *reference, long n, const uint8_t *query, long m) {
    for (long r = n - m; r >= 0; --r) {
        long q = 0;
        while (q < m && reference[r + q] == query[q]) ++q;
        if (q == m) return r;
    }
    return -1;
}

/* :124-271.  sw and btrack are (n+1) x (m+1), zero-initialised by the caller (Array2::zeros). */
static void calculate_matrix(const uint8_t *reference, long rl, const uint8_t *alternate, long al, int32_t *sw,
                             int32_t *btrack, int strategy, int32_t w_match, int32_t w_mismatch, int32_t w_open,
                             int32_t w_extend) {
    const long nrow = rl + 1, ncol = al + 1;
    const int32_t low_init_value = INT32_MIN / 2;
    int32_t *best_gap_v = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ncol + 1));
    int32_t *gap_size_v = (int32_t *)calloc((size_t)(ncol + 1), sizeof(int32_t));
    int32_t *best_gap_h = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nrow + 1));
    int32_t *gap_size_h = (int32_t *)calloc((size_t)(nrow + 1), sizeof(int32_t));
    for (long j = 0; j <= ncol; ++j) best_gap_v[j] = low_init_value;
    for (long i = 0; i <= nrow; ++i) best_gap_h[i] = low_init_value;
#define SW(i, j) sw[(size_t)(i) * (size_t)ncol + (size_t)(j)]
#define BT(i, j) btrack[(size_t)(i) * (size_t)ncol + (size_t)(j)]
    if (strategy == SW_INDEL || strategy == SW_LEADING_INDEL) { /* :145-178 */
        int32_t current_value = w_open;
        SW(0, 1) = current_value;
        for (long j = 2; j < ncol; ++j) {
            current_value += w_extend;
            SW(0, j) = current_value;
        }
        SW(1, 0) = w_open;
        current_value = w_open;
        for (long i = 2; i < nrow; ++i) {
            current_value += w_extend;
            SW(i, 0) = current_value;
        }
    }
    for (long i = 1; i < nrow; ++i) { /* :190-270 */
        const uint8_t a_base = reference[i - 1];
        for (long j = 1; j < ncol; ++j) {
            const uint8_t b_base = alternate[j - 1];
            const int32_t step_diag = SW(i - 1, j - 1) + (a_base == b_base ? w_match : w_mismatch);
            int32_t prev_gap = SW(i - 1, j) + w_open;
            best_gap_v[j] += w_extend;
            if (prev_gap > best_gap_v[j]) {
                best_gap_v[j] = prev_gap;
                gap_size_v[j] = 1;
            } else {
                gap_size_v[j] += 1;
            }
            const int32_t step_down = best_gap_v[j];
            const int32_t kd = gap_size_v[j];
            prev_gap = SW(i, j - 1) + w_open;
            best_gap_h[i] += w_extend;
            if (prev_gap > best_gap_h[i]) {
                best_gap_h[i] = prev_gap;
                gap_size_h[i] = 1;
            } else {
                gap_size_h[i] += 1;
            }
            const int32_t step_right = best_gap_h[i];
            const int32_t ki = gap_size_h[i];
            const int diag_highest_or_equal = step_diag >= step_down && step_diag >= step_right;
            if (diag_highest_or_equal) {
                SW(i, j) = step_diag > MATRIX_MIN_CUTOFF ? step_diag : MATRIX_MIN_CUTOFF;
                BT(i, j) = 0;
            } else if (step_right >= step_down) {
                SW(i, j) = step_right > MATRIX_MIN_CUTOFF ? step_right : MATRIX_MIN_CUTOFF;
                BT(i, j) = -ki;
            } else {
                SW(i, j) = step_down > MATRIX_MIN_CUTOFF ? step_down : MATRIX_MIN_CUTOFF;
                BT(i, j) = kd;
            }
        }
    }
    free(best_gap_v);
    free(gap_size_v);
    free(best_gap_h);
    free(gap_size_h);
}

/* :273-443.  Writes the elements front to back into cigar[], returns their number. */
static uint32_t calculate_cigar(const int32_t *sw, const int32_t *btrack, long ref_length, long alt_length, int strategy,
                                uint32_t *cigar, int32_t *alignment_offset_out) {
    const long ncol = alt_length + 1;
    long p1 = 0, p2;
    int32_t max_score = INT32_MIN;
    int32_t segment_length = 0;
    if (strategy == SW_INDEL) {
        p1 = ref_length;
        p2 = alt_length;
    } else {
        p2 = alt_length;
        for (long i = 1; i <= ref_length; ++i) {
            const int32_t cur_score = SW(i, alt_length);
            if (cur_score >= max_score) {
                p1 = i;
                max_score = cur_score;
            }
        }
        if (strategy != SW_LEADING_INDEL) {
            for (long j = 1; j <= alt_length; ++j) {
                const int32_t cur_score = SW(ref_length, j);
                if (cur_score > max_score ||
                    (cur_score == max_score && labs(ref_length - j) < labs(p1 - p2))) {
                    p1 = ref_length;
                    p2 = j;
                    max_score = cur_score;
                    segment_length = (int32_t)(alt_length - j);
                }
            }
        }
    }
    uint32_t n = 0; /* lce, built backwards */
    if (segment_length > 0 && strategy == SW_SOFTCLIP) {
        cigar[n++] = make_element(ST_CLIP, (uint32_t)segment_length);
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
        switch (new_state) {
            case ST_MATCH: p1 -= 1; p2 -= 1; break;
            case ST_INSERTION: p2 -= step_length; break;
            case ST_DELETION: p1 -= step_length; break;
        }
        if (new_state == state) {
            segment_length += step_length;
        } else {
            if (segment_length > 0) cigar[n++] = make_element(state, (uint32_t)segment_length);
            segment_length = step_length;
            state = new_state;
        }
        if (p1 <= 0 || p2 <= 0) break;
    }
    int32_t alignment_offset;
    if (strategy == SW_SOFTCLIP) {
        cigar[n++] = make_element(state, (uint32_t)segment_length);
        if (p2 > 0) cigar[n++] = make_element(ST_CLIP, (uint32_t)p2);
        alignment_offset = (int32_t)p1;
    } else if (strategy == SW_IGNORE) {
        cigar[n++] = make_element(state, (uint32_t)(segment_length + p2));
        alignment_offset = (int32_t)(p1 - p2);
    } else {
        cigar[n++] = make_element(state, (uint32_t)segment_length);
        if (p1 > 0)
            cigar[n++] = make_element(ST_DELETION, (uint32_t)p1);
        else if (p2 > 0)
            cigar[n++] = make_element(ST_INSERTION, (uint32_t)p2);
        alignment_offset = 0;
    }
    for (uint32_t a = 0, b = n ? n - 1 : 0; a < b; ++a, --b) { /* lce.reverse() */
        const uint32_t t = cigar[a];
        cigar[a] = cigar[b];
        cigar[b] = t;
    }
    *alignment_offset_out = alignment_offset;
    return n;
}
#undef SW
#undef BT

/* :47-107, AVXMode::None arm.  cigar must hold ref_len + alt_len + 3 elements ((len << 4) | BAM op).
 * Returns the number of elements, or -1 for empty input (the reference asserts / panics, :65-68,132-134). */
ORACLE_API int oracle_sw_align(const uint8_t *reference, uint32_t ref_len, const uint8_t *alternate, uint32_t alt_len,
                               int32_t w_match, int32_t w_mismatch, int32_t w_open, int32_t w_extend, int strategy,
                               uint32_t *cigar, int32_t *alignment_offset) {
    if (ref_len == 0 || alt_len == 0) return -1;
    if (strategy == SW_SOFTCLIP || strategy == SW_IGNORE) {
        const long match_index = last_index_of(reference, ref_len, alternate, alt_len);
        if (match_index >= 0) {
            cigar[0] = make_element(ST_MATCH, alt_len);
            *alignment_offset = (int32_t)match_index;
            return 1;
        }
    }
    const size_t cells = ((size_t)ref_len + 1) * ((size_t)alt_len + 1);
    int32_t *sw = (int32_t *)calloc(cells, sizeof(int32_t));
    int32_t *btrack = (int32_t *)calloc(cells, sizeof(int32_t));
    calculate_matrix(reference, ref_len, alternate, alt_len, sw, btrack, strategy, w_match, w_mismatch, w_open, w_extend);
    const uint32_t n = calculate_cigar(sw, btrack, ref_len, alt_len, strategy, cigar, alignment_offset);
    free(sw);
    free(btrack);
    return (int)n;
}

/* Batch driver with the SoA layout of phmm_sw_align (include/phmm.h): alignment a owns cigar[cigar_off[a] .. cigar_off[a+1]). */
ORACLE_API int oracle_sw_align_batch(uint32_t n_alignments, const uint32_t *ref_off, const uint8_t *ref_bases,
                                     const uint32_t *alt_off, const uint8_t *alt_bases, int32_t w_match,
                                     int32_t w_mismatch, int32_t w_open, int32_t w_extend, int strategy,
                                     const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar,
                                     int32_t *alignment_offset) {
    for (uint32_t a = 0; a < n_alignments; ++a) {
        const uint32_t rl = ref_off[a + 1] - ref_off[a], al = alt_off[a + 1] - alt_off[a];
        uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)rl + al + 3));
        const int n = oracle_sw_align(ref_bases + ref_off[a], rl, alt_bases + alt_off[a], al, w_match, w_mismatch, w_open,
                                      w_extend, strategy, tmp, &alignment_offset[a]);
        if (n < 0) {
            free(tmp);
            return -1;
        }
        n_cigar[a] = (uint32_t)n;
        const uint64_t cap = cigar_off[a + 1] - cigar_off[a];
        memcpy(cigar + cigar_off[a], tmp, sizeof(uint32_t) * (size_t)((uint64_t)n < cap ? (uint64_t)n : cap));
        free(tmp);
    }
    return 0;
}
