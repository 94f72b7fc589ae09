/*
 * pairhmm_oracle.c -- CPU ORACLE for the PairHMM read x haplotype log10-likelihood path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (lorikeet_amd/, the HIP
 * kernels behind include/phmm.h) never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the *scalar* ("logless", AVXMode::None) arm of the
 * reference, following it statement by statement so that it can be pinned against the
 * reference's own known-answer vectors.  Each function cites the reference file:line.
 *
 * Parity pinning: the restatement is checked (tests/test_oracle.py) against all 104 vectors
 * of the reference fixture tests/resources/pairhmm-testdata.txt (committed as data under
 * tests/golden/) with the reference's own tolerance, 1e-5 absolute
 * (tests/vector_pair_hmm_unit_tests.rs:63,90), and against the analytic expectations of
 * tests/pair_hmm_unit_tests.rs and tests/pair_hmm_model_unit_tests.rs.
 *
 * What cannot be checked here: the reference's production arithmetic in AVX mode is the
 * third-party crate `gkl ^0.1.1` (Cargo.toml:42, no Cargo.lock => no exact pin), whose source
 * is not under /root/reference, and no Rust toolchain exists in this image, so the reference
 * itself can be neither compiled nor run.  gkl's published algorithm is the Intel GKL PairHMM:
 * the same M/I/D forward recurrence evaluated in f32 with a 2^120 initial scale and re-run in
 * f64 (2^1020) when the f32 result underflows; it is pinned to this scalar recurrence by the
 * 104-vector fixture at 1e-5.  This oracle restates the scalar recurrence (f64, 2^1020).
 *
 * Arithmetic notes:
 *   - Rust never contracts a*b+c into an FMA; build this file with -ffp-contract=off (the
 *     Makefile does) so the operation order and rounding match the Rust source.
 *   - `10.0_f64.powf(x)` is restated as pow(10.0, x).  (LLVM may lower the Rust call to
 *     exp10(x); glibc's pow and exp10 can differ in the last ulp -- ~1e-16 relative, eleven
 *     orders of magnitude below the 1e-5 gate.)
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * QualityUtils -- src/utils/quality_utils.rs
 * ---------------------------------------------------------------------------------------- */
#define MAX_QUAL 254          /* quality_utils.rs:24 */
#define MIN_USABLE_Q_SCORE 6  /* quality_utils.rs:23 */

/* quality_utils.rs:98-104  qual_to_error_prob: 10^(q / -10) */
ORACLE_API double oracle_qual_to_error_prob(uint8_t qual) { return pow(10.0, ((double)qual) / -10.0); }

/* quality_utils.rs:82-88  qual_to_prob: 1 - qual_to_error_prob */
ORACLE_API double oracle_qual_to_prob(uint8_t qual) { return 1.0 - oracle_qual_to_error_prob(qual); }

/* ------------------------------------------------------------------------------------------
 * MathUtils::approximate_log10_sum_log10 + JacobianLogTable -- src/utils/math_utils.rs
 * ---------------------------------------------------------------------------------------- */
#define JLT_MAX_TOLERANCE 8.0   /* math_utils.rs:485 */
#define JLT_TABLE_STEP 0.0001   /* math_utils.rs:490 */

static double *jlt_cache = NULL; /* math_utils.rs:9-14 (lazy_static cache) */
static size_t jlt_cache_len = 0;
static pthread_once_t jlt_once = PTHREAD_ONCE_INIT;

static void jlt_build(void) {
    /* math_utils.rs:9-14: (0..((MAX_TOLERANCE / TABLE_STEP) + 1.0) as usize)
     *                       .map(|k| (1.0 + 10^(-(k as f64) * TABLE_STEP)).log10()) */
    size_t n = (size_t)((JLT_MAX_TOLERANCE / JLT_TABLE_STEP) + 1.0);
    double *c = (double *)malloc(n * sizeof(double));
    for (size_t k = 0; k < n; ++k) c[k] = log10(1.0 + pow(10.0, -((double)k) * JLT_TABLE_STEP));
    jlt_cache = c;
    jlt_cache_len = n;
}

/* math_utils.rs:493-497  JacobianLogTable::get: cache[round(diff * INV_STEP)] */
static double jlt_get(double difference) {
    const double inv_step = 1.0 / JLT_TABLE_STEP; /* math_utils.rs:491 */
    size_t index = (size_t)round(difference * inv_step); /* f64::round == C round (half away from 0) */
    return jlt_cache[index];
}

/* math_utils.rs:314-332 */
ORACLE_API double oracle_approximate_log10_sum_log10(double a, double b) {
    pthread_once(&jlt_once, jlt_build);
    if (a > b) return oracle_approximate_log10_sum_log10(b, a);
    if (a == -INFINITY) return b;
    double diff = b - a;
    return b + (diff < JLT_MAX_TOLERANCE ? jlt_get(diff) : 0.0);
}

/* ------------------------------------------------------------------------------------------
 * PairHMMModel -- src/pair_hmm/pair_hmm_model.rs
 * ---------------------------------------------------------------------------------------- */
enum { /* pair_hmm_model.rs:83-113 */
    T_MATCH_TO_MATCH = 0,
    T_INDEL_TO_MATCH = 1,
    T_MATCH_TO_INSERTION = 2,
    T_INSERTION_TO_INSERTION = 3,
    T_MATCH_TO_DELETION = 4,
    T_DELETION_TO_DELETION = 5,
    TRANS_PROB_ARRAY_LENGTH = 6
};

#define MM_TABLE_LEN (((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1) /* pair_hmm_model.rs:48-53 */
static double *mm_prob_table = NULL;  /* match_to_match_prob  */
static double *mm_log10_table = NULL; /* match_to_match_log10 */
static pthread_once_t mm_once = PTHREAD_ONCE_INIT;

/* pair_hmm_model.rs:47-78  PairHMMModel::new (process-global here; the reference rebuilds it per
 * PairHMM, i.e. per region -- same values every time). */
static void mm_build(void) {
    const double inv_ln10 = 1.0 / log(10.0); /* pair_hmm_model.rs:13,18 */
    double *p = (double *)malloc(MM_TABLE_LEN * sizeof(double));
    double *l = (double *)malloc(MM_TABLE_LEN * sizeof(double));
    size_t offset = 0;
    for (int i = 0; i <= MAX_QUAL; ++i) {
        for (int j = 0; j <= i; ++j) {
            double log10_sum = oracle_approximate_log10_sum_log10(-0.1 * (double)i, -0.1 * (double)j);
            double log10_sum_pow = pow(10.0, log10_sum);
            double m = log10_sum_pow < 1.0 ? log10_sum_pow : 1.0; /* min(1.0, pow) */
            l[offset + j] = log1p(-m) * inv_ln10;
            p[offset + j] = pow(10.0, l[offset + j]);
        }
        offset += (size_t)i + 1;
    }
    mm_prob_table = p;
    mm_log10_table = l;
}

/* pair_hmm_model.rs:442-461  match_to_match_prob */
ORACLE_API double oracle_match_to_match_prob(unsigned ins_qual, unsigned del_qual) {
    pthread_once(&mm_once, mm_build);
    unsigned min_qual, max_qual;
    if (ins_qual <= del_qual) {
        min_qual = ins_qual;
        max_qual = del_qual;
    } else {
        min_qual = del_qual;
        max_qual = ins_qual;
    }
    if ((unsigned)MAX_QUAL < max_qual) {
        return 1.0 - pow(10.0, oracle_approximate_log10_sum_log10(-0.1 * (double)min_qual, -0.1 * (double)max_qual));
    }
    return mm_prob_table[((max_qual * (max_qual + 1)) >> 1) + min_qual];
}

/* pair_hmm_model.rs:142-156  qual_to_trans_probs_with_array1 */
ORACLE_API void oracle_qual_to_trans_probs(double *dest, uint8_t ins_qual, uint8_t del_qual, uint8_t gcp) {
    dest[T_MATCH_TO_MATCH] = oracle_match_to_match_prob(ins_qual, del_qual);
    dest[T_MATCH_TO_INSERTION] = oracle_qual_to_error_prob(ins_qual);
    dest[T_MATCH_TO_DELETION] = oracle_qual_to_error_prob(del_qual);
    dest[T_INDEL_TO_MATCH] = oracle_qual_to_prob(gcp);
    double tmp = oracle_qual_to_error_prob(gcp);
    dest[T_INSERTION_TO_INSERTION] = tmp;
    dest[T_DELETION_TO_DELETION] = tmp;
}

/* ------------------------------------------------------------------------------------------
 * PairHMM (scalar arm) -- src/pair_hmm/pair_hmm.rs
 * ---------------------------------------------------------------------------------------- */
typedef struct oracle_pairhmm {
    /* pair_hmm.rs:25-50 (fields used by the scalar arm) */
    int constants_are_initialized;
    long previous_haplotype_length; /* -1 == None */
    long hap_start_index;           /* -1 == None */
    size_t max_haplotype_length, max_read_length;
    size_t padded_max_read_length, padded_max_haplotype_length;
    size_t padded_read_length, padded_haplotype_length;
    int initialized;
    int do_not_use_tristate_correction;
    double *transition; /* (max_read_length+1) x 6            */
    double *prior;      /* padded_max_read x padded_max_hap   */
    double *match_matrix, *insertion_matrix, *deletion_matrix;
} oracle_pairhmm;

#define IDX(h, i, j) ((size_t)(i) * (h)->padded_max_haplotype_length + (size_t)(j))

/* pair_hmm.rs:128-165  quick_initialize */
ORACLE_API oracle_pairhmm *oracle_pairhmm_new(size_t max_read_length, size_t haplotype_max_length) {
    oracle_pairhmm *h = (oracle_pairhmm *)calloc(1, sizeof(*h));
    h->max_read_length = max_read_length;
    h->max_haplotype_length = haplotype_max_length;
    h->padded_max_read_length = max_read_length + 1;
    h->padded_max_haplotype_length = haplotype_max_length + 1;
    h->initialized = !(max_read_length == 0 && haplotype_max_length == 0);
    size_t n = h->padded_max_read_length * h->padded_max_haplotype_length;
    h->match_matrix = (double *)calloc(n, sizeof(double));
    h->insertion_matrix = (double *)calloc(n, sizeof(double));
    h->deletion_matrix = (double *)calloc(n, sizeof(double));
    h->prior = (double *)calloc(n, sizeof(double));
    h->transition = (double *)calloc((max_read_length + 1) * TRANS_PROB_ARRAY_LENGTH, sizeof(double));
    h->previous_haplotype_length = -1;
    h->hap_start_index = -1;
    return h;
}

ORACLE_API void oracle_pairhmm_free(oracle_pairhmm *h) {
    if (!h) return;
    free(h->match_matrix);
    free(h->insertion_matrix);
    free(h->deletion_matrix);
    free(h->prior);
    free(h->transition);
    free(h);
}

/* pair_hmm.rs:189-191 */
ORACLE_API void oracle_pairhmm_do_not_use_tristate_correction(oracle_pairhmm *h) {
    h->do_not_use_tristate_correction = 1;
}

/* pair_hmm.rs:706-717 */
ORACLE_API size_t oracle_find_first_position_where_haplotypes_differ(const uint8_t *h1, size_t n1,
                                                                     const uint8_t *h2, size_t n2) {
    size_t n = n1 < n2 ? n1 : n2;
    for (size_t i = 0; i < n; ++i)
        if (h1[i] != h2[i]) return i;
    return n;
}

/* pair_hmm.rs:626-673  initialize_priors.  The reference visits the whole (max-size) prior
 * matrix and writes only cells with 0<i<=R and start_index<j<=H. */
static void initialize_priors(oracle_pairhmm *h, const uint8_t *hap, size_t H, const uint8_t *read, size_t R,
                              const uint8_t *quals, size_t start_index) {
    const double TRISTATE_CORRECTION = 3.0; /* pair_hmm.rs:53 */
    for (size_t i = 1; i <= R; ++i) {
        uint8_t x = read[i - 1];
        uint8_t qual = quals[i - 1];
        for (size_t j = start_index + 1; j <= H; ++j) {
            uint8_t y = hap[j - 1];
            h->prior[IDX(h, i, j)] =
                (x == y || x == 'N' || y == 'N')
                    ? oracle_qual_to_prob(qual)
                    : (oracle_qual_to_error_prob(qual) / (h->do_not_use_tristate_correction ? 1.0 : TRISTATE_CORRECTION));
        }
    }
}

/* pair_hmm.rs:682-694 -> pair_hmm_model.rs:197-232: rows 1..=R of the transition matrix */
static void initialize_probabilities(oracle_pairhmm *h, const uint8_t *ins, const uint8_t *del, const uint8_t *gcp,
                                     size_t R) {
    for (size_t i = 0; i < R; ++i)
        oracle_qual_to_trans_probs(&h->transition[(i + 1) * TRANS_PROB_ARRAY_LENGTH], ins[i], del[i], gcp[i]);
}

/* pair_hmm.rs:503-615  sub_compute_read_likelihood_given_haplotype_log10 */
static double sub_compute(oracle_pairhmm *h, const uint8_t *hap, size_t H, const uint8_t *read, size_t R,
                          const uint8_t *quals, const uint8_t *ins, const uint8_t *del, const uint8_t *gcp,
                          size_t hap_start_index, int recache_read_values) {
    const double INITIAL_CONDITION = pow(2.0, 1020.0);      /* pair_hmm.rs:16 */
    const double INITIAL_CONDITION_LOG10 = log10(INITIAL_CONDITION); /* pair_hmm.rs:17 */

    /* :515-529  first row of the deletion matrix (whole allocated row, like row_mut(0).fill) */
    if (h->previous_haplotype_length < 0 || (size_t)h->previous_haplotype_length != H) {
        double initial_value = INITIAL_CONDITION / (double)H;
        for (size_t j = 0; j < h->padded_max_haplotype_length; ++j) h->deletion_matrix[IDX(h, 0, j)] = initial_value;
    }
    /* :531-534 */
    if (!h->constants_are_initialized || recache_read_values) {
        initialize_probabilities(h, ins, del, gcp, R);
        h->constants_are_initialized = 1;
    }
    /* :536 */
    initialize_priors(h, hap, H, read, R, quals, hap_start_index);

    /* :573-593  the hot loop */
    for (size_t i = 1; i < h->padded_read_length; ++i) {
        const double *t = &h->transition[i * TRANS_PROB_ARRAY_LENGTH];
        for (size_t j = hap_start_index + 1; j < h->padded_haplotype_length; ++j) {
            h->match_matrix[IDX(h, i, j)] =
                h->prior[IDX(h, i, j)] * (h->match_matrix[IDX(h, i - 1, j - 1)] * t[T_MATCH_TO_MATCH] +
                                          h->insertion_matrix[IDX(h, i - 1, j - 1)] * t[T_INDEL_TO_MATCH] +
                                          h->deletion_matrix[IDX(h, i - 1, j - 1)] * t[T_INDEL_TO_MATCH]);
            h->insertion_matrix[IDX(h, i, j)] = h->match_matrix[IDX(h, i - 1, j)] * t[T_MATCH_TO_INSERTION] +
                                                h->insertion_matrix[IDX(h, i - 1, j)] * t[T_INSERTION_TO_INSERTION];
            h->deletion_matrix[IDX(h, i, j)] = h->match_matrix[IDX(h, i, j - 1)] * t[T_MATCH_TO_DELETION] +
                                               h->deletion_matrix[IDX(h, i, j - 1)] * t[T_DELETION_TO_DELETION];
        }
    }

    /* :598-614  sum of the last row of M and I, then back to log10 */
    size_t end_i = h->padded_read_length - 1;
    double final_sum_probabilities = 0.0;
    for (size_t j = 1; j < h->padded_haplotype_length; ++j)
        final_sum_probabilities += h->match_matrix[IDX(h, end_i, j)] + h->insertion_matrix[IDX(h, end_i, j)];
    return log10(final_sum_probabilities) - INITIAL_CONDITION_LOG10;
}

/* pair_hmm.rs:405-501  compute_read_likelihood_given_haplotype_log10.
 * next_hap == NULL <=> None.  Returns NaN and sets *status (if non-NULL) on the conditions the
 * reference asserts on (:417-440, :478-481): 1 not initialized, 2 haplotype too long,
 * 3 array length mismatch (not representable here: one length R for all), 4 result > 0. */
ORACLE_API double oracle_compute_read_likelihood_given_haplotype_log10(
    oracle_pairhmm *h, const uint8_t *hap, size_t H, const uint8_t *read, size_t R, const uint8_t *quals,
    const uint8_t *ins, const uint8_t *del, const uint8_t *gcp, int recache_read_values, const uint8_t *next_hap,
    size_t next_H, int *status) {
    if (status) *status = 0;
    if (!h->initialized) {
        if (status) *status = 1;
        return NAN;
    }
    if (H > h->max_haplotype_length) {
        if (status) *status = 2;
        return NAN;
    }
    h->padded_read_length = R + 1;
    h->padded_haplotype_length = H + 1;
    if (recache_read_values) h->hap_start_index = 0; /* :444-448 */

    /* :452-464 */
    size_t next_hap_start_index = 0;
    if (next_hap != NULL && H == next_H)
        next_hap_start_index = oracle_find_first_position_where_haplotypes_differ(hap, H, next_hap, next_H);

    double result =
        sub_compute(h, hap, H, read, R, quals, ins, del, gcp, (size_t)h->hap_start_index, recache_read_values);

    if (!(result <= 0.0) && status) *status = 4; /* :478-481 */

    /* :487-496 */
    if (h->hap_start_index < 0)
        h->hap_start_index = (long)next_hap_start_index;
    else
        h->hap_start_index = ((long)next_hap_start_index < h->hap_start_index) ? 0 : (long)next_hap_start_index;

    h->previous_haplotype_length = (long)H; /* :498 */
    return result;
}

/* ------------------------------------------------------------------------------------------
 * One region, scalar arm of compute_log10_likelihoods -- pair_hmm.rs:268-338, one call per
 * sample; here all reads of the region are handed in one list (results do not depend on the
 * sample split: each (read, haplotype) value is independent).  A fresh PairHMM per region
 * (pair_hmm_likelihood_calculation_engine.rs:212), sized by PairHMM::initialize (:110-122).
 * Output: out[r * n_haps + a], read-major, haplotypes in list order (pair_hmm.rs:289-337).
 * ---------------------------------------------------------------------------------------- */
static int region_compute(uint32_t n_reads, uint32_t n_haps, const uint32_t *read_off, const uint8_t *read_bases,
                          const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp,
                          const uint32_t *hap_off, const uint8_t *hap_bases, int disable_tristate, double *out) {
    if (n_reads == 0) return 0; /* pair_hmm.rs:224 */
    size_t max_r = 0, max_h = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        size_t len = read_off[r + 1] - read_off[r];
        if (len > max_r) max_r = len;
    }
    for (uint32_t a = 0; a < n_haps; ++a) {
        size_t len = hap_off[a + 1] - hap_off[a];
        if (len > max_h) max_h = len;
    }
    oracle_pairhmm *h = oracle_pairhmm_new(max_r, max_h);
    if (disable_tristate) oracle_pairhmm_do_not_use_tristate_correction(h);
    int worst = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        size_t ro = read_off[r], R = read_off[r + 1] - ro;
        int is_first_haplotype = 1;
        for (uint32_t a = 0; a < n_haps; ++a) {
            size_t ho = hap_off[a], H = hap_off[a + 1] - ho;
            const uint8_t *next = NULL;
            size_t next_H = 0;
            if (a + 1 < n_haps) {
                next = hap_bases + hap_off[a + 1];
                next_H = hap_off[a + 2] - hap_off[a + 1];
            }
            int st = 0;
            out[(size_t)r * n_haps + a] = oracle_compute_read_likelihood_given_haplotype_log10(
                h, hap_bases + ho, H, read_bases + ro, R, base_q + ro, ins_q + ro, del_q + ro, gcp + ro,
                is_first_haplotype, next, next_H, &st);
            if (st > worst) worst = st;
            is_first_haplotype = 0;
        }
    }
    oracle_pairhmm_free(h);
    return worst;
}

/* Batch driver with the same SoA layout as phmm_compute (include/phmm.h).  Regions are handed to
 * `n_threads` pthreads from a shared counter, one region per task -- the reference's
 * rayon-over-regions scheme (src/assembly/assembly_region_walker.rs:210-273). */
typedef struct {
    uint32_t n_regions;
    const uint32_t *region_read_off, *region_hap_off, *read_off, *hap_off;
    const uint8_t *read_bases, *base_q, *ins_q, *del_q, *gcp, *hap_bases;
    const uint64_t *out_off;
    double *out;
    int disable_tristate;
    volatile uint32_t next;
    volatile int status;
} batch_ctx;

static void *batch_worker(void *p) {
    batch_ctx *c = (batch_ctx *)p;
    for (;;) {
        uint32_t g = __sync_fetch_and_add(&c->next, 1u);
        if (g >= c->n_regions) break;
        uint32_t r0 = c->region_read_off[g], r1 = c->region_read_off[g + 1];
        uint32_t h0 = c->region_hap_off[g], h1 = c->region_hap_off[g + 1];
        int st = region_compute(r1 - r0, h1 - h0, c->read_off + r0, c->read_bases, c->base_q, c->ins_q, c->del_q,
                                c->gcp, c->hap_off + h0, c->hap_bases, c->disable_tristate, c->out + c->out_off[g]);
        if (st) c->status = st;
    }
    return NULL;
}

ORACLE_API int oracle_compute(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                              const uint32_t *read_off, const uint8_t *read_bases, const uint8_t *base_q,
                              const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp, const uint32_t *hap_off,
                              const uint8_t *hap_bases, const uint64_t *out_off, double *out, int disable_tristate,
                              int n_threads) {
    pthread_once(&jlt_once, jlt_build);
    pthread_once(&mm_once, mm_build);
    batch_ctx c = {n_regions, region_read_off, region_hap_off, read_off, hap_off, read_bases, base_q, ins_q,
                   del_q,     gcp,             hap_bases,      out_off,  out,     disable_tristate, 0, 0};
    if (n_threads <= 1) {
        batch_worker(&c);
        return c.status;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, batch_worker, &c);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    free(th);
    return c.status;
}

/* Table export so tests can compare the product's host-built device tables with the oracle's. */
ORACLE_API size_t oracle_mm_table_len(void) { return MM_TABLE_LEN; }
ORACLE_API const double *oracle_mm_prob_table(void) {
    pthread_once(&jlt_once, jlt_build);
    pthread_once(&mm_once, mm_build);
    return mm_prob_table;
}
