/* TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu baseline): never linked into or called by
 * lorikeet_amd/.
 *
 * CPU restatement, statement by statement, of the CIGAR algebra that projects a read's alignment to its best haplotype
 * onto the reference -- everything AlignmentUtils::create_read_aligned_to_ref does after the Smith-Waterman call:
 *   src/reads/cigar_builder.rs:43-372        CigarBuilder (add, add_all, make, make_and_record_deletions_removed_result)
 *   src/reads/cigar_utils.rs:105-127,459-464,492-603,649-666   element predicates and constructors
 *   src/haplotype/haplotype.rs:248-256       Haplotype::get_consolidated_padded_cigar
 *   src/reads/alignment_utils.rs:60-165      create_read_aligned_to_ref (from the alignment on)
 *   src/reads/alignment_utils.rs:173-213     append_clipped_elements_from_cigar_to_cigar
 *   src/reads/alignment_utils.rs:240-281     apply_cigar_to_cigar   (:974-1061 CigarPairTransform)
 *   src/reads/alignment_utils.rs:283-311     read_start_on_reference_haplotype
 *   src/reads/alignment_utils.rs:321-402     trim_cigar_by_bases / trim_cigar
 *   src/reads/alignment_utils.rs:425-566     left_align_indels      (:585-678 normalize_alleles)
 * CIGAR elements travel in BAM encoding, (length << 4) | op with M 0, I 1, D 2, N 3, S 4, H 5, P 6, = 7, X 8.
 * Pinned by the data of the reference's own tests (tests/alignment_utils_unit_tests.rs, tests/cigar_builder_unit_tests.rs)
 * restated in tests/test_cigar_oracle.py.  Where the reference panics or returns Err, these functions return a negative
 * status. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };
enum {
    CIG_OK = 0,
    CIG_ERR_RIGHT_CLIP = -1,      /* "Cigar has already reached its right (hard) clip" */
    CIG_ERR_ALL_SOFT_CLIPPED = -2, /* "Cigar is completely soft clipped" */
    CIG_ERR_LAST_NONE = -3,       /* "Last element cannot be None at this point" */
    CIG_ERR_EMPTY = -4,           /* "No cigar elements left after removing leading and trailing deletions." */
    CIG_ERR_PANIC = -5,           /* an assert! / panic! / arithmetic overflow of the reference */
    CIG_ERR_CAPACITY = -6         /* more elements than this restatement reserves */
};

#define CAP 8192
typedef uint32_t elem_t;
static uint32_t e_len(elem_t e) { return e >> 4; }
static int e_op(elem_t e) { return (int)(e & 15u); }
static elem_t mk(int op, uint32_t len) { return (len << 4) | (uint32_t)op; }

/* cigar_utils.rs:105-127, :459-464, :536-541, :661-666 */
static int consumes_read(elem_t e) { const int o = e_op(e); return o == OP_M || o == OP_EQ || o == OP_X || o == OP_I || o == OP_S; }
static int consumes_ref(elem_t e) { const int o = e_op(e); return o == OP_M || o == OP_D || o == OP_N || o == OP_EQ || o == OP_X; }
static int is_clipping(elem_t e) { return e_op(e) == OP_S || e_op(e) == OP_H; }
static int is_indel(elem_t e) { return e_op(e) == OP_D || e_op(e) == OP_I; }
static int is_alignment(elem_t e) { return e_op(e) == OP_M || e_op(e) == OP_EQ || e_op(e) == OP_X; }
static uint32_t length_on_read(elem_t e) { return consumes_read(e) ? e_len(e) : 0; }  /* alignment_utils.rs:388-394 */
static uint32_t length_on_reference(elem_t e) { return consumes_ref(e) ? e_len(e) : 0; } /* :396-402 */

/* ---- CigarBuilder (cigar_builder.rs) ------------------------------------------------------------------------------- */
enum { SEC_LEFT_HARD, SEC_LEFT_SOFT, SEC_MIDDLE, SEC_RIGHT_SOFT, SEC_RIGHT_HARD };
typedef struct {
    elem_t el[CAP];
    size_t n;
    int has_last, last_op; /* last_operator: only its operator is ever consulted */
    int section, remove_deletions_at_ends;
    uint32_t leading_removed, trailing_removed, trailing_removed_in_make;
    int error;
} builder_t;

static void builder_init(builder_t *b, int remove_deletions_at_ends) { /* :46-57 */
    memset(b, 0, sizeof *b);
    b->remove_deletions_at_ends = remove_deletions_at_ends;
    b->section = SEC_LEFT_HARD;
}

static int last_two_were_deletion_and_insertion(const builder_t *b) { /* :201-220 */
    return b->has_last && b->n > 1 && b->last_op == OP_I && e_op(b->el[b->n - 2]) == OP_D;
}

static int advance_section(builder_t *b, elem_t e) { /* :223-273 */
    const int op = e_op(e);
    if (op == OP_H) {
        if (b->section == SEC_LEFT_SOFT || b->section == SEC_MIDDLE || b->section == SEC_RIGHT_SOFT) b->section = SEC_RIGHT_HARD;
    } else if (op == OP_S) {
        if (b->section == SEC_RIGHT_HARD) return b->error = CIG_ERR_RIGHT_CLIP;
        if (b->section == SEC_LEFT_HARD) b->section = SEC_LEFT_SOFT;
        else if (b->section == SEC_MIDDLE) b->section = SEC_RIGHT_SOFT;
    } else {
        if (b->section == SEC_RIGHT_SOFT || b->section == SEC_RIGHT_HARD) return b->error = CIG_ERR_RIGHT_CLIP;
        if (b->section == SEC_LEFT_HARD || b->section == SEC_LEFT_SOFT) b->section = SEC_MIDDLE;
    }
    return CIG_OK;
}

static int push(builder_t *b, elem_t e) {
    if (b->n >= CAP) return CIG_ERR_CAPACITY;
    b->el[b->n++] = e;
    return CIG_OK;
}

static int builder_add(builder_t *b, elem_t element) { /* :59-186 */
    if (e_len(element) == 0) return CIG_OK;
    if (b->remove_deletions_at_ends && e_op(element) == OP_D) { /* :61-86: a deletion at the start of the alignment is dropped */
        int leading;
        if (!b->has_last) leading = 1;
        else if (b->last_op == OP_S || b->last_op == OP_H) leading = 1;
        else if (b->last_op == OP_I) leading = b->n == 1 || is_clipping(b->el[b->n - 2]);
        else leading = 0;
        if (leading) {
            b->leading_removed += e_len(element);
            return CIG_OK;
        }
    }
    const int st = advance_section(b, element); /* :88-91 */
    if (st != CIG_OK) return st;
    if (b->has_last && b->last_op == e_op(element)) { /* :93-97 cigar_elements_are_same_type: merge */
        const size_t n = b->n - 1;
        if (e_op(b->el[n]) == e_op(element)) b->el[n] = mk(e_op(element), e_len(element) + e_len(b->el[n])); /* combine or keep */
        return CIG_OK;
    }
    if (!b->has_last) { /* :99-103 */
        b->has_last = 1;
        b->last_op = e_op(element);
        return push(b, element);
    }
    if (is_clipping(element)) { /* :105-131 */
        const size_t len = b->n;
        const elem_t last = mk(b->last_op, 1);
        if (b->remove_deletions_at_ends && !consumes_read(last) && !is_clipping(last)) {
            /* clipping starts on the right and the last operator was a deletion: the clip replaces it */
            b->trailing_removed += e_len(b->el[len - 1]);
            b->el[len - 1] = element;
            b->last_op = e_op(element);
            return CIG_OK;
        }
        if (b->remove_deletions_at_ends && last_two_were_deletion_and_insertion(b)) {
            /* ... or deletion + insertion: the deletion goes (last_operator stays the insertion, as in the reference) */
            b->trailing_removed += e_len(b->el[len - 2]);
            b->el[len - 2] = b->el[len - 1];
            b->el[len - 1] = element;
            return CIG_OK;
        }
        b->last_op = e_op(element);
        return push(b, element);
    }
    if (e_op(element) == OP_D && b->last_op == OP_I) { /* :133-170: deletions move to the left of an adjacent insertion */
        const size_t size = b->n;
        if (size > 1 && e_op(b->el[size - 2]) == OP_D) {
            b->el[size - 2] = mk(OP_D, e_len(b->el[size - 2]) + e_len(element));
        } else {
            if (b->n >= CAP) return CIG_ERR_CAPACITY;
            b->el[size] = b->el[size - 1];
            b->el[size - 1] = element;
            b->n++;
        }
        return CIG_OK; /* last_operator remains the insertion */
    }
    b->last_op = e_op(element);
    return push(b, element);
}

static int builder_add_all(builder_t *b, const elem_t *e, size_t n) { /* :188-199 */
    for (size_t i = 0; i < n; ++i) {
        const int st = builder_add(b, e[i]);
        if (st != CIG_OK) return st == CIG_ERR_CAPACITY ? st : CIG_ERR_RIGHT_CLIP;
    }
    return CIG_OK;
}

static int builder_make(builder_t *b, int allow_empty) { /* :275-324; the elements stay in b->el */
    if (b->error != CIG_OK) return b->error;
    if (b->section == SEC_LEFT_SOFT && b->n && e_op(b->el[0]) == OP_S) return CIG_ERR_ALL_SOFT_CLIPPED;
    b->trailing_removed_in_make = 0;
    if (b->remove_deletions_at_ends) {
        if (!b->has_last) return CIG_ERR_LAST_NONE;
        if (b->last_op == OP_D) {
            b->trailing_removed_in_make = e_len(b->el[b->n - 1]);
            b->n--;
        } else if (last_two_were_deletion_and_insertion(b)) {
            b->trailing_removed_in_make = e_len(b->el[b->n - 2]);
            b->el[b->n - 2] = b->el[b->n - 1];
            b->n--;
        }
    }
    if (!allow_empty && b->n == 0) return CIG_ERR_EMPTY;
    return CIG_OK;
}

/* make_and_record_deletions_removed_result (:326-339): leading / trailing deletion bases that were dropped */
static int builder_make_result(builder_t *b, uint32_t *leading, uint32_t *trailing) {
    const int st = builder_make(b, 0);
    if (st != CIG_OK) return st;
    *leading = b->leading_removed;
    *trailing = b->trailing_removed + b->trailing_removed_in_make;
    return CIG_OK;
}

static int copy_out(const builder_t *b, elem_t *out, size_t cap, uint32_t *n_out) {
    *n_out = (uint32_t)b->n;
    if (b->n > cap) return CIG_ERR_CAPACITY;
    memcpy(out, b->el, b->n * sizeof(elem_t));
    return CIG_OK;
}

/* test entry: add_all + make(allow_empty) */
ORACLE_API int oracle_cigar_builder(const uint32_t *elements, uint32_t n, int remove_deletions_at_ends, int allow_empty,
                                    uint32_t *out, uint32_t cap, uint32_t *n_out, uint32_t *leading_removed,
                                    uint32_t *trailing_removed) {
    builder_t *b = (builder_t *)malloc(sizeof *b);
    builder_init(b, remove_deletions_at_ends);
    int st = CIG_OK;
    for (uint32_t i = 0; i < n && st == CIG_OK; ++i) st = builder_add(b, elements[i]);
    if (st == CIG_OK) st = builder_make(b, allow_empty);
    if (st == CIG_OK) {
        *leading_removed = b->leading_removed;
        *trailing_removed = b->trailing_removed + b->trailing_removed_in_make;
        st = copy_out(b, out, cap, n_out);
    }
    free(b);
    return st;
}

/* Haplotype::get_consolidated_padded_cigar (haplotype.rs:248-256) */
static int consolidated_padded_cigar(const elem_t *cigar, size_t n, uint32_t pad, builder_t *b) {
    builder_init(b, 1);
    int st = builder_add_all(b, cigar, n);
    if (st == CIG_OK) st = builder_add(b, mk(OP_M, pad));
    if (st == CIG_OK) st = builder_make(b, 0);
    return st;
}

/* ... as the reference's test calls it (tests/haplotype_unit_tests.rs:96-146; tests/golden/consolidate_cigar_cases.json) */
ORACLE_API int oracle_consolidated_padded_cigar(const uint32_t *cigar, uint32_t n, uint32_t pad, uint32_t *out, uint32_t cap, uint32_t *n_out) {
    builder_t *b = (builder_t *)malloc(sizeof *b);
    if (!b) return CIG_ERR_PANIC;
    int st = consolidated_padded_cigar(cigar, n, pad, b);
    if (st == CIG_OK) st = copy_out(b, out, cap, n_out);
    free(b);
    return st;
}

/* ---- alignment_utils.rs -------------------------------------------------------------------------------------------- */
/* :283-311 */
static int read_start_on_reference_haplotype(const elem_t *cigar, size_t n, uint32_t read_start_on_haplotype, uint32_t *out) {
    if (read_start_on_haplotype == 0) {
        *out = 0;
        return CIG_OK;
    }
    uint32_t ref_consumed = 0, hap_consumed = 0;
    for (size_t i = 0; i < n; ++i) {
        ref_consumed += length_on_reference(cigar[i]);
        hap_consumed += length_on_read(cigar[i]);
        if (hap_consumed >= read_start_on_haplotype) {
            const uint32_t excess = consumes_ref(cigar[i]) ? hap_consumed - read_start_on_haplotype : 0; /* saturating_sub: never negative here */
            *out = ref_consumed >= excess ? ref_consumed - excess : 0;
            return CIG_OK;
        }
    }
    return CIG_ERR_PANIC; /* "Cigar doesn't reach the read start" */
}

ORACLE_API int oracle_read_start_on_reference_haplotype(const uint32_t *cigar, uint32_t n, uint32_t start, uint32_t *out) {
    return read_start_on_reference_haplotype(cigar, n, start, out);
}

/* :334-386 */
static int trim_cigar(const elem_t *cigar, size_t n, uint32_t start, uint32_t end, int by_reference, builder_t *b, uint32_t *leading,
                      uint32_t *trailing) {
    if (end < start) return CIG_ERR_PANIC; /* "End position cannot be before start position" */
    builder_init(b, 1);
    uint64_t element_start, element_end = 0;
    for (size_t i = 0; i < n; ++i) {
        const elem_t elt = cigar[i];
        element_start = element_end;
        element_end = element_start + (by_reference ? length_on_reference(elt) : length_on_read(elt));
        /* zero-length elements at both ends are included: elementStart == elementEnd == start, or == end + 1 */
        if (element_end < start || (element_end == start && element_start < start)) continue;
        if (element_start > end && element_end > (uint64_t)end + 1) break;
        int64_t overlap;
        if (element_end == element_start) overlap = e_len(elt);
        else overlap = (int64_t)((uint64_t)end + 1 < element_end ? (uint64_t)end + 1 : element_end) - (int64_t)(start > element_start ? start : element_start);
        if (overlap < 0) return CIG_ERR_PANIC; /* u32 underflow */
        const int st = builder_add(b, mk(e_op(elt), (uint32_t)overlap));
        if (st != CIG_OK) return st;
    }
    if (element_end < end) return CIG_ERR_PANIC; /* "Cigar elements don't reach end position (inclusive)" */
    return builder_make_result(b, leading, trailing);
}

ORACLE_API int oracle_trim_cigar(const uint32_t *cigar, uint32_t n, uint32_t start, uint32_t end, int by_reference, uint32_t *out,
                                 uint32_t cap, uint32_t *n_out, uint32_t *leading, uint32_t *trailing) {
    builder_t *b = (builder_t *)malloc(sizeof *b);
    int st = trim_cigar(cigar, n, start, end, by_reference, b, leading, trailing);
    if (st == CIG_OK) st = copy_out(b, out, cap, n_out);
    free(b);
    return st;
}

/* CigarPairTransform::new (:974-1049): op13 (-1 = none), advance12, advance23 */
static int pair_transform(elem_t op12, elem_t op23, int *op13, uint32_t *adv12, uint32_t *adv23) {
    const int a = e_op(op12), c = e_op(op23);
    const int a_match = a == OP_M || a == OP_EQ || a == OP_X, a_ins = a == OP_I || a == OP_S, a_del = a == OP_D;
    const int c_match = c == OP_M || c == OP_EQ || c == OP_X, c_ins = c == OP_I || c == OP_S, c_del = c == OP_D;
    if (!(c_match || c_ins || c_del)) return CIG_ERR_PANIC;
    if (a_match) {
        if (c_match) { *op13 = OP_M; *adv12 = 1; *adv23 = 1; }
        else if (c_ins) { *op13 = OP_I; *adv12 = 1; *adv23 = 1; }
        else { *op13 = OP_D; *adv12 = 0; *adv23 = 1; }
    } else if (a_ins) {
        *op13 = OP_I; *adv12 = 1; *adv23 = 0;
    } else if (a_del) {
        if (c_match) { *op13 = OP_D; *adv12 = 1; *adv23 = 1; }
        else if (c_ins) { *op13 = -1; *adv12 = 1; *adv23 = 1; }
        else { *op13 = OP_D; *adv12 = 0; *adv23 = 1; }
    } else {
        return CIG_ERR_PANIC;
    }
    return CIG_OK;
}

/* :240-281 */
static int apply_cigar_to_cigar(const elem_t *c12, size_t n12, const elem_t *c23, size_t n23, builder_t *b) {
    builder_init(b, 1);
    size_t i12 = 0, i23 = 0;
    uint32_t e12 = 0, e23 = 0;
    while (i12 < n12 && i23 < n23) {
        int op13;
        uint32_t a12, a23;
        const int st = pair_transform(c12[i12], c23[i23], &op13, &a12, &a23);
        if (st != CIG_OK) return st;
        e12 += a12;
        e23 += a23;
        if (op13 >= 0) {
            const int s2 = builder_add(b, mk(op13, 1));
            if (s2 != CIG_OK) return s2 == CIG_ERR_CAPACITY ? s2 : CIG_ERR_PANIC; /* .expect("Failed to add cigar element") */
        }
        if (e12 == e_len(c12[i12])) { /* the current element is used up */
            ++i12;
            e12 = 0;
        }
        if (e23 == e_len(c23[i23])) {
            ++i23;
            e23 = 0;
        }
    }
    const int st = builder_make(b, 0);
    return st == CIG_OK ? st : (st == CIG_ERR_CAPACITY ? st : CIG_ERR_PANIC);
}

ORACLE_API int oracle_apply_cigar_to_cigar(const uint32_t *c12, uint32_t n12, const uint32_t *c23, uint32_t n23, uint32_t *out,
                                           uint32_t cap, uint32_t *n_out) {
    builder_t *b = (builder_t *)malloc(sizeof *b);
    int st = apply_cigar_to_cigar(c12, n12, c23, n23, b);
    if (st == CIG_OK) st = copy_out(b, out, cap, n_out);
    free(b);
    return st;
}

/* normalize_alleles (:585-640) for the two sequences left_align_indels passes (reference, read), trim = true */
typedef struct { int32_t start, end; } range_t;
static uint8_t base_at(const uint8_t *seq, size_t len, int64_t i, int *ok) {
    if (i < 0 || (uint64_t)i >= len) {
        *ok = 0;
        return 0;
    }
    return seq[i];
}
static int32_t r_len(range_t r) { return r.end > r.start ? r.end - r.start : 0; } /* Range::len() of an empty / inverted range is 0 */

static int normalize_alleles(const uint8_t *seq0, size_t len0, const uint8_t *seq1, size_t len1, range_t *b0, range_t *b1,
                             uint32_t max_shift, int trim, int32_t *start_shift_out, int32_t *end_shift_out) {
    if (max_shift > (uint32_t)b0->start || max_shift > (uint32_t)b1->start) return CIG_ERR_PANIC; /* "maxShift goes past the start of a sequence" (`bound.start as u32`) */
#define AT(seq, len, i) base_at(seq, len, (int64_t)(i), &ok)
    int ok = 1; /* cleared by an index outside a sequence: the reference panics there */
    int32_t start_shift = 0, end_shift = 0;
    int32_t min_size = r_len(*b0) < r_len(*b1) ? r_len(*b0) : r_len(*b1);
    /* consume any redundant shared bases at the end of the alleles */
    while (trim && min_size > 0 && AT(seq0, len0, b0->end - 1) == AT(seq1, len1, b1->end - 1)) {
        if (!ok) return CIG_ERR_PANIC;
        b0->end -= 1;
        b1->end -= 1;
        min_size -= 1;
        end_shift += 1;
    }
    while (trim && min_size > 0 && AT(seq0, len0, b0->start) == AT(seq1, len1, b1->start)) {
        if (!ok) return CIG_ERR_PANIC;
        b0->start += 1;
        b1->start += 1;
        min_size -= 1;
        start_shift -= 1;
    }
    /* shift left as long as the last bases on the right are equal among all sequences and the next bases on the left are */
    while (start_shift < (int32_t)max_shift && AT(seq0, len0, b0->start - 1) == AT(seq1, len1, b1->start - 1) &&
           AT(seq0, len0, b0->end - 1) == AT(seq1, len1, b1->end - 1)) {
        if (!ok) return CIG_ERR_PANIC;
        b0->start -= 1;
        b0->end -= 1;
        b1->start -= 1;
        b1->end -= 1;
        start_shift += 1;
        end_shift += 1;
    }
    if (!ok) return CIG_ERR_PANIC; /* an index outside a sequence panics in the reference */
#undef AT
    *start_shift_out = start_shift;
    *end_shift_out = end_shift;
    return CIG_OK;
}

/* :425-566 */
static int left_align_indels(const elem_t *cigar, size_t n, const uint8_t *ref_seq, size_t ref_len, const uint8_t *read, size_t read_len,
                             uint32_t read_start, builder_t *b, uint32_t *leading, uint32_t *trailing) {
    int any_indel = 0;
    size_t last_indel = 0;
    for (size_t i = 0; i < n; ++i)
        if (is_indel(cigar[i])) {
            any_indel = 1;
            last_indel = i;
        }
    if (!any_indel) { /* :431-433: the cigar as it is */
        if (n > CAP) return CIG_ERR_CAPACITY;
        builder_init(b, 1);
        memcpy(b->el, cigar, n * sizeof(elem_t));
        b->n = n;
        *leading = *trailing = 0;
        return CIG_OK;
    }
    /* we need reference bases from the start of the read to the rightmost indel */
    uint64_t necessary = read_start;
    for (size_t i = 0; i <= last_indel; ++i) necessary += length_on_reference(cigar[i]);
    if (necessary > ref_len) return CIG_ERR_PANIC; /* "Read goes past end of reference" */

    static __thread elem_t rtl[CAP]; /* result_right_to_left */
    size_t n_rtl = 0;
#define RTL(e)                                    \
    do {                                          \
        if (n_rtl >= CAP) return CIG_ERR_CAPACITY; \
        rtl[n_rtl++] = (e);                       \
    } while (0)
    uint32_t ref_length = 0;
    for (size_t i = 0; i < n; ++i) ref_length += length_on_reference(cigar[i]);
    range_t ref_r = {(int32_t)(read_start + ref_length), (int32_t)(read_start + ref_length)};
    range_t read_r = {(int32_t)read_len, (int32_t)read_len};
    for (size_t k = n; k-- > 0;) {
        const elem_t element = cigar[k];
        if (is_indel(element)) { /* accumulate; the shift happens when an alignment block or the read start is reached */
            ref_r.start -= (int32_t)length_on_reference(element);
            read_r.start -= (int32_t)length_on_read(element);
        } else if (r_len(ref_r) == 0 && r_len(read_r) == 0) {
            ref_r.start -= (int32_t)length_on_reference(element);
            read_r.start -= (int32_t)length_on_read(element);
            ref_r.end -= (int32_t)length_on_reference(element);
            read_r.end -= (int32_t)length_on_read(element);
            RTL(element);
        } else {
            const uint32_t max_shift = is_alignment(element) ? e_len(element) : 0;
            int32_t shift0, shift1;
            const int st = normalize_alleles(ref_seq, ref_len, read, read_len, &ref_r, &read_r, max_shift, 1, &shift0, &shift1);
            if (st != CIG_OK) return st;
            RTL(mk(OP_M, (uint32_t)shift1)); /* new match alignments on the right due to left-alignment */
            /* emit if we didn't go all the way to the start of an alignment block OR we have reached clips OR the start */
            const int emit_indel = k == 0 || shift0 < (int32_t)max_shift || !is_alignment(element);
            const int32_t new_match_left = shift0 < 0 ? -shift0 : 0;
            const int32_t remaining_left = shift0 < 0 ? (int32_t)e_len(element) : (int32_t)e_len(element) - shift0;
            if (emit_indel) {
                RTL(mk(OP_D, (uint32_t)r_len(ref_r)));
                RTL(mk(OP_I, (uint32_t)r_len(read_r)));
                ref_r.end -= r_len(ref_r);   /* now empty, pointing at the start of the left-aligned indel */
                read_r.end -= r_len(read_r);
                const int32_t dref = new_match_left + (consumes_ref(element) ? remaining_left : 0);
                const int32_t dread = new_match_left + (consumes_read(element) ? remaining_left : 0);
                ref_r.start -= dref;
                ref_r.end -= dref;
                read_r.start -= dread;
                read_r.end -= dread;
            }
            RTL(mk(OP_M, (uint32_t)new_match_left));
            if (remaining_left < 0) return CIG_ERR_PANIC; /* `as u32` of a negative length */
            RTL(mk(e_op(element), (uint32_t)remaining_left));
        }
    }
    RTL(mk(OP_D, (uint32_t)r_len(ref_r)));
    RTL(mk(OP_I, (uint32_t)r_len(read_r)));
#undef RTL
    if (read_r.start != 0) return CIG_ERR_PANIC; /* "Given cigar does not account for all bases of the read" */
    builder_init(b, 1);
    for (size_t i = n_rtl; i-- > 0;) {
        const int st = builder_add(b, rtl[i]);
        if (st != CIG_OK) return st == CIG_ERR_CAPACITY ? st : CIG_ERR_RIGHT_CLIP; /* add_all(...)? */
    }
    return builder_make_result(b, leading, trailing);
}

ORACLE_API int oracle_left_align_indels(const uint32_t *cigar, uint32_t n, const uint8_t *ref_seq, uint32_t ref_len, const uint8_t *read,
                                        uint32_t read_len, uint32_t read_start, uint32_t *out, uint32_t cap, uint32_t *n_out,
                                        uint32_t *leading, uint32_t *trailing) {
    builder_t *b = (builder_t *)malloc(sizeof *b);
    int st = left_align_indels(cigar, n, ref_seq, ref_len, read, read_len, read_start, b, leading, trailing);
    if (st == CIG_OK) st = copy_out(b, out, cap, n_out);
    free(b);
    return st;
}

/* :173-213 */
static int append_clipped_elements(const elem_t *cigar, size_t n, const elem_t *original, size_t n_original, elem_t *out, size_t cap,
                                   uint32_t *n_out) {
    if (n_original == 0) return CIG_ERR_PANIC; /* indexing [0] of an empty cigar */
    size_t first = 0, last = n_original - 1, m = 0;
#define OUT(e)                                  \
    do {                                        \
        if (m >= cap) return CIG_ERR_CAPACITY;  \
        out[m++] = (e);                         \
    } while (0)
    while (is_clipping(original[first]) && first != last) {
        OUT(original[first]);
        ++first;
    }
    for (size_t i = 0; i < n; ++i) OUT(cigar[i]);
    /* the clips on the right, kept in their original order (soft before hard) */
    size_t right_begin = last + 1;
    while (is_clipping(original[last]) && first != last) {
        right_begin = last;
        --last;
    }
    for (size_t i = right_begin; i < n_original; ++i) OUT(original[i]);
#undef OUT
    *n_out = (uint32_t)m;
    return CIG_OK;
}

ORACLE_API int oracle_append_clipped_elements(const uint32_t *cigar, uint32_t n, const uint32_t *original, uint32_t n_original,
                                              uint32_t *out, uint32_t cap, uint32_t *n_out) {
    return append_clipped_elements(cigar, n, original, n_original, out, cap, n_out);
}

/* create_read_aligned_to_ref from the alignment on (:60-165).
 *   sw_cigar / sw_offset          the read -> haplotype alignment (SoftClip, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS)
 *   hap_cigar, hap_start_wrt_ref  Haplotype::cigar and alignment_start_hap_wrt_ref of the best haplotype
 *   reference_start               padded_reference_loc.get_start()
 *   ref_bases                     the reference haplotype's bases
 *   read                          the read minus its soft clips (what was aligned)
 *   original_cigar                the read's cigar before realignment (for its clips); original_read_len its length
 * Returns 1 when the read is returned unchanged (alignment_offset == -1, :60-63), 0 with the new position and cigar,
 * or a negative status where the reference panics. */
ORACLE_API int oracle_create_read_aligned_to_ref(const uint32_t *sw_cigar, uint32_t n_sw, int32_t sw_offset, const uint32_t *hap_cigar,
                                                 uint32_t n_hap_cigar, uint32_t hap_start_wrt_ref, uint64_t reference_start,
                                                 const uint8_t *ref_bases, uint32_t ref_len, const uint8_t *read, uint32_t read_len,
                                                 const uint32_t *original_cigar, uint32_t n_original, uint32_t original_read_len,
                                                 int64_t *new_pos, uint32_t *out, uint32_t cap, uint32_t *n_out) {
    if (sw_offset == -1) return 1;
    if (sw_offset < 0) return CIG_ERR_PANIC;
    builder_t *b = (builder_t *)malloc(3 * sizeof *b), *b2 = b + 1, *b3 = b + 2;
    int st;
    uint32_t lead = 0, trail = 0;
#define TRY(x)              \
    do {                    \
        st = (x);           \
        if (st != CIG_OK) { \
            free(b);        \
            return st < 0 ? st : CIG_ERR_PANIC; \
        }                   \
    } while (0)
    /* :65-72 sw_cigar = CigarBuilder(true).add_all(alignment cigar).make(false) */
    builder_init(b, 1);
    TRY(builder_add_all(b, sw_cigar, n_sw));
    TRY(builder_make(b, 0));
    /* :84-100 */
    TRY(consolidated_padded_cigar(hap_cigar, n_hap_cigar, 1000, b2));
    uint32_t start_on_ref_hap = 0;
    TRY(read_start_on_reference_haplotype(b2->el, b2->n, (uint32_t)sw_offset, &start_on_ref_hap));
    const uint64_t read_start_on_reference = reference_start + hap_start_wrt_ref + start_on_ref_hap;
    /* :107-113: the haplotype -> reference cigar from the read start on (elements after the read end are kept) */
    uint32_t padded_read_length = 0;
    for (size_t i = 0; i < b2->n; ++i) padded_read_length += length_on_read(b2->el[i]);
    if (padded_read_length == 0) { free(b); return CIG_ERR_PANIC; }
    TRY(trim_cigar(b2->el, b2->n, (uint32_t)sw_offset, padded_read_length - 1, 0, b3, &lead, &trail));
    /* :115-122 */
    TRY(apply_cigar_to_cigar(b->el, b->n, b3->el, b3->n, b2));
    TRY(left_align_indels(b2->el, b2->n, ref_bases, ref_len, read, read_len, start_on_ref_hap, b3, &lead, &trail));
    /* :126-130: left-alignment may have moved a deletion to the front of the read and removed it */
    *new_pos = (int64_t)(read_start_on_reference + lead);
    /* :135-143 */
    TRY(append_clipped_elements(b3->el, b3->n, original_cigar, n_original, out, cap, n_out));
    /* :151-161 */
    uint32_t aligned_read_len = 0;
    for (size_t i = 0; i < b3->n; ++i) aligned_read_len += length_on_read(b3->el[i]);
    const uint32_t soft_clipped_bases = original_read_len - read_len;
    st = aligned_read_len + soft_clipped_bases == original_read_len ? 0 : CIG_ERR_PANIC;
#undef TRY
    free(b);
    return st;
}

/* ---- CigarUtils::calculate_cigar (src/reads/cigar_utils.rs:358-457): the CIGAR of a haplotype against the reference --------
 * Smith-Waterman between the two sequences, each padded with SW_PAD ("NNNNNNNNNN", :11) on both sides, the padding
 * trimmed off again, indels left-aligned, and leading / trailing deletions -- which the builder strips -- put back so that
 * the reference span stays.  Returns 0 with the cigar, 1 for None (is_s_w_failure, :469-487), negative where the
 * reference panics.  The aligner is the other oracle file's (oracle_sw_align). */
int oracle_sw_align(const uint8_t *reference, uint32_t ref_len, const uint8_t *alternate, uint32_t alt_len, int32_t w_match,
                    int32_t w_mismatch, int32_t w_open, int32_t w_extend, int strategy, uint32_t *cigar, int32_t *alignment_offset);

ORACLE_API int oracle_calculate_cigar(const uint8_t *ref_seq, uint32_t ref_len, const uint8_t *alt_seq, uint32_t alt_len, int32_t w_match,
                                      int32_t w_mismatch, int32_t w_open, int32_t w_extend, int strategy, uint32_t *out, uint32_t cap,
                                      uint32_t *n_out) {
    if (cap < 1) return CIG_ERR_CAPACITY;
    if (alt_len == 0) { /* :365-368 "horrible edge case from the unit tests, where this path has no bases" */
        out[0] = mk(OP_D, ref_len);
        *n_out = 1;
        return CIG_OK;
    }
    if (alt_len == ref_len) { /* :370-385: equal lengths and at most two mismatches: all M */
        uint32_t mismatches = 0;
        for (uint32_t i = 0; i < ref_len; ++i) mismatches += alt_seq[i] != ref_seq[i];
        if (mismatches <= 2) {
            out[0] = mk(OP_M, ref_len);
            *n_out = 1;
            return CIG_OK;
        }
    }
    enum { PAD = 10 };
    const uint32_t pr = ref_len + 2 * PAD, pa = alt_len + 2 * PAD;
    uint8_t *padded_ref = (uint8_t *)malloc(pr), *padded_alt = (uint8_t *)malloc(pa);
    memset(padded_ref, 'N', pr);
    memset(padded_alt, 'N', pa);
    memcpy(padded_ref + PAD, ref_seq, ref_len);
    memcpy(padded_alt + PAD, alt_seq, alt_len);
    uint32_t *sw = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)pr + pa + 4));
    int32_t offset = 0;
    const int n_sw = oracle_sw_align(padded_ref, pr, padded_alt, pa, w_match, w_mismatch, w_open, w_extend, strategy, sw, &offset);
    free(padded_ref);
    free(padded_alt);
    int st = CIG_OK;
    builder_t *b = (builder_t *)malloc(2 * sizeof *b), *b2 = b + 1;
    uint32_t lead = 0, trail = 0, lead2 = 0, trail2 = 0;
    if (n_sw < 0) st = CIG_ERR_PANIC;
    if (st == CIG_OK && offset > 0) st = 1; /* is_s_w_failure: the alignment must start at the first base ... */
    for (int i = 0; st == CIG_OK && i < n_sw; ++i)
        if (e_op(sw[i]) == OP_S) st = 1;    /* ... and have no S operators */
    /* :421-428 cut off the padding bases */
    if (st == CIG_OK) st = trim_cigar(sw, (size_t)n_sw, PAD, pa - PAD - 1, 0, b, &lead, &trail);
    if (st == CIG_OK && trail > 0) { /* :430-435 the trailing deletion goes back on for the left-alignment */
        if (b->n >= CAP) st = CIG_ERR_CAPACITY;
        else b->el[b->n++] = mk(OP_D, trail);
    }
    if (st == CIG_OK) st = left_align_indels(b->el, b->n, ref_seq, ref_len, alt_seq, alt_len, lead, b2, &lead2, &trail2);
    if (st == CIG_OK) { /* :444-466 */
        const uint32_t total_leading = lead + lead2, total_trailing = trail2;
        uint32_t m = 0;
        if ((size_t)b2->n + 2 > cap) st = CIG_ERR_CAPACITY;
        else {
            if (total_leading > 0) out[m++] = mk(OP_D, total_leading);
            for (size_t i = 0; i < b2->n; ++i) out[m++] = b2->el[i];
            if (total_trailing > 0) out[m++] = mk(OP_D, total_trailing);
            *n_out = m;
        }
    }
    free(sw);
    free(b);
    return st;
}
