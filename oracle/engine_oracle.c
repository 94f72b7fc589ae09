/*
 * engine_oracle.c -- CPU ORACLE for the engine-level steps around the PairHMM kernel.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as pairhmm_oracle.c: tests/, smoke() and bench.py's
 * cpu_baseline leg only; never linked or called by lorikeet_amd/).
 *
 * Plain-C restatement of
 *   PairHMMLikelihoodCalculationEngine   src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs
 *     initialize_pcr_error_model / get_error_model_adjusted_qual   :169-193
 *     modify_read_qualities (modify_soft_clipped_bases branch)      :352-388
 *     cap_minimum_read_qualities                                    :428-466
 *     apply_pcr_error_model / find_tandem_repeat_units              :502-611
 *     log10_min_true_likelihood / dynamic read-qual threshold       :244-319 (table :23-39)
 *   VariantContextUtils::find_number_of_repetitions(_main)          src/model/variant_context_utils.rs:240-335
 *   AlleleLikelihoods::normalize_likelihoods / search_best_allele   src/model/allele_likelihoods.rs:378-554
 *   AlleleLikelihoods::filter_poorly_modeled_evidence               src/model/allele_likelihoods.rs:925-1041
 *
 * Pinned by the known answers SURVEY.md section 4 derives for the reference's own engine fixture
 * (tests/pair_hmm_likelihood_calculation_engine_unit_tests.rs:21-88): PCR cache, the read's ins/del
 * quals [38 x 9, 45], raw log10 L(ref) / L(alt), the normalised alt value and the static threshold
 * (tests/test_engine_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

#define MAX_STR_UNIT_LENGTH 20 /* engine.rs:98 */
#define MAX_REPEAT_LENGTH 100  /* engine.rs:99 */
#define MIN_ADJUSTED_QSCORE 6  /* engine.rs:100 */
#define INITIAL_QSCORE 40.0    /* engine.rs:105 */
#define MIN_USABLE_Q_SCORE 6   /* quality_utils.rs:23 */

/* engine.rs:186-193  get_error_model_adjusted_qual.  Rust `as usize` saturates negatives to 0. */
static uint8_t error_model_adjusted_qual(size_t repeat_length, double rate_factor) {
    double v = INITIAL_QSCORE - exp((double)repeat_length / (rate_factor * M_PI)) + 1.0;
    size_t u = (v > 0.0) ? (size_t)v : 0;
    if (v != v) u = 0;
    size_t m = u > MIN_ADJUSTED_QSCORE ? u : MIN_ADJUSTED_QSCORE;
    return (uint8_t)m; /* `as u8` truncates; m <= 41 here */
}

/* engine.rs:169-184  initialize_pcr_error_model.  model: 0 None, 1 Hostile, 2 Aggressive, 3 Conservative
 * (engine.rs:61-70; the enum discriminant IS the rate factor). */
ORACLE_API void oracle_pcr_error_model_cache(int model, uint8_t cache[MAX_REPEAT_LENGTH + 1]) {
    memset(cache, 0, MAX_REPEAT_LENGTH + 1);
    if (model == 0) return;
    for (size_t i = 0; i <= MAX_REPEAT_LENGTH; ++i) cache[i] = error_model_adjusted_qual(i, (double)model);
}

/* variant_context_utils.rs:276-335  find_number_of_repetitions_main */
static size_t find_number_of_repetitions_main(const uint8_t *repeat_unit_full, size_t offset_in_repeat_unit_full,
                                              size_t repeat_unit_length, const uint8_t *test_string_full,
                                              size_t offset_in_test_string_full, size_t test_string_length,
                                              int leading_repeats) {
    if (test_string_length == 0) return 0;
    long length_difference = (long)test_string_length - (long)repeat_unit_length;
    size_t num_repeats = 0;
    if (leading_repeats) {
        /* (0..=length_difference).step_by(repeat_unit_length) */
        for (long start = 0; start <= length_difference; start += (long)repeat_unit_length) {
            if (memcmp(test_string_full + start + offset_in_test_string_full,
                       repeat_unit_full + offset_in_repeat_unit_full, repeat_unit_length) == 0)
                num_repeats += 1;
            else
                return num_repeats;
        }
        return num_repeats;
    }
    /* (0..=length_difference).rev().step_by(repeat_unit_length): length_difference, -unit, ... >= 0 */
    for (long start = length_difference; start >= 0; start -= (long)repeat_unit_length) {
        if (memcmp(test_string_full + start + offset_in_test_string_full, repeat_unit_full + offset_in_repeat_unit_full,
                   repeat_unit_length) == 0)
            num_repeats += 1;
        else
            return num_repeats;
    }
    return num_repeats;
}

/* variant_context_utils.rs:240-257 */
static size_t find_number_of_repetitions(const uint8_t *repeat_unit, size_t unit_len, const uint8_t *test_string,
                                         size_t test_len, int leading_repeats) {
    if (test_len == 0) return 0;
    return find_number_of_repetitions_main(repeat_unit, 0, unit_len, test_string, 0, test_len, leading_repeats);
}

/* the two functions above as the reference's tests call them (tests/variant_context_utils_unit_tests.rs:23-294), so that the
 * restatement is pinned by those assertions (tests/golden/repetition_cases.json) and not only through their caller below */
ORACLE_API size_t oracle_find_number_of_repetitions(const uint8_t *repeat_unit, size_t unit_len, const uint8_t *test_string,
                                                    size_t test_len, int leading_repeats) {
    return find_number_of_repetitions(repeat_unit, unit_len, test_string, test_len, leading_repeats);
}
ORACLE_API size_t oracle_find_number_of_repetitions_main(const uint8_t *repeat_unit_full, size_t offset_in_repeat_unit_full,
                                                         size_t repeat_unit_length, const uint8_t *test_string_full,
                                                         size_t offset_in_test_string_full, size_t test_string_length,
                                                         int leading_repeats) {
    return find_number_of_repetitions_main(repeat_unit_full, offset_in_repeat_unit_full, repeat_unit_length, test_string_full,
                                           offset_in_test_string_full, test_string_length, leading_repeats);
}

/* engine.rs:528-611  find_tandem_repeat_units -> repeat length (the unit itself is not used by the caller) */
ORACLE_API size_t oracle_find_tandem_repeat_length(const uint8_t *read_bases, size_t n, size_t offset) {
    size_t max_bw = 0;
    const uint8_t *best_bw_unit = read_bases + offset;
    size_t best_bw_len = 1;
    for (size_t str = 1; str <= MAX_STR_UNIT_LENGTH; ++str) {
        if (offset + 1 < str) break; /* (offset + 1).checked_sub(str).is_none() */
        max_bw = find_number_of_repetitions_main(read_bases, offset + 1 - str, str, read_bases, 0, offset + 1, 0);
        if (max_bw > 1) {
            best_bw_unit = read_bases + (offset + 1 - str);
            best_bw_len = str;
            break;
        }
    }
    const uint8_t *best_unit = best_bw_unit;
    size_t best_len = best_bw_len;
    size_t max_rl = max_bw;

    if (offset < n - 1) {
        const uint8_t *best_fw_unit = read_bases + offset + 1;
        size_t best_fw_len = 1;
        size_t max_fw = 0;
        for (size_t str = 1; str <= MAX_STR_UNIT_LENGTH; ++str) {
            if (offset + str + 1 > n) break;
            max_fw = find_number_of_repetitions_main(read_bases, offset + 1, str, read_bases, offset + 1, n - offset - 1, 1);
            if (max_fw > 1) {
                best_fw_unit = read_bases + offset + 1;
                best_fw_len = str;
                break;
            }
        }
        if (best_fw_len == best_len && memcmp(best_fw_unit, best_unit, best_len) == 0) {
            max_rl = max_bw + max_fw;
        } else {
            max_bw = find_number_of_repetitions(best_fw_unit, best_fw_len, read_bases, offset + 1, 0);
            max_rl = max_fw + max_bw;
        }
    }
    if (max_rl > MAX_REPEAT_LENGTH) max_rl = MAX_REPEAT_LENGTH;
    return max_rl;
}

/* engine.rs:502-526  apply_pcr_error_model (all but the last base) */
ORACLE_API void oracle_apply_pcr_error_model(int model, const uint8_t *read_bases, size_t n, uint8_t *ins, uint8_t *del) {
    if (model == 0) return;
    uint8_t cache[MAX_REPEAT_LENGTH + 1];
    oracle_pcr_error_model_cache(model, cache);
    for (size_t i = 1; i < n; ++i) {
        size_t repeat_length = oracle_find_tandem_repeat_length(read_bases, n, i - 1);
        if (cache[repeat_length] < ins[i - 1]) ins[i - 1] = cache[repeat_length];
        if (cache[repeat_length] < del[i - 1]) del[i - 1] = cache[repeat_length];
    }
}

/* engine.rs:428-466  cap_minimum_read_qualities + set_to_fixed_value_if_too_low */
ORACLE_API void oracle_cap_minimum_read_qualities(uint8_t mapq, uint8_t *quals, uint8_t *ins, uint8_t *del, size_t n,
                                                  uint8_t base_quality_score_threshold,
                                                  int disable_cap_read_qualities_to_mapq) {
    for (size_t i = 0; i < n; ++i) {
        if (!disable_cap_read_qualities_to_mapq && mapq < quals[i]) quals[i] = mapq;
        if (quals[i] < base_quality_score_threshold) quals[i] = MIN_USABLE_Q_SCORE;
        if (ins[i] < MIN_USABLE_Q_SCORE) ins[i] = MIN_USABLE_Q_SCORE;
        if (del[i] < MIN_USABLE_Q_SCORE) del[i] = MIN_USABLE_Q_SCORE;
    }
}

/* engine.rs:352-388  modify_read_qualities for one read, in place (default branch). */
ORACLE_API void oracle_modify_read_qualities(int model, const uint8_t *read_bases, size_t n, uint8_t mapq, uint8_t *quals,
                                             uint8_t *ins, uint8_t *del, uint8_t base_quality_score_threshold,
                                             int disable_cap_read_qualities_to_mapq) {
    oracle_apply_pcr_error_model(model, read_bases, n, ins, del);
    oracle_cap_minimum_read_qualities(mapq, quals, ins, del, n, base_quality_score_threshold,
                                      disable_cap_read_qualities_to_mapq);
}

/* engine.rs:23-39  dynamic_read_qual_thresh_lookup_table: baseQ, mean, variance */
static const double dynamic_read_qual_thresh_lookup_table[] = {
    1.0,  5.996842844, 0.196616587, 2.0,  5.870018422, 1.388545569, 3.0,  5.401558531, 5.641990128,
    4.0,  4.818940919, 10.33176216, 5.0,  4.218758304, 14.25799688, 6.0,  3.646319832, 17.02880749,
    7.0,  3.122346753, 18.64537883, 8.0,  2.654731979, 19.27521677, 9.0,  2.244479156, 19.13584613,
    10.0, 1.88893867,  18.43922003, 11.0, 1.583645342, 17.36842261, 12.0, 1.3233807,   16.07088712,
    13.0, 1.102785365, 14.65952563, 14.0, 0.916703025, 13.21718577, 15.0, 0.760361881, 11.80207947,
    16.0, 0.629457387, 10.45304833, 17.0, 0.520175654, 9.194183767, 18.0, 0.42918208,  8.038657241,
    19.0, 0.353590663, 6.991779595, 20.0, 0.290923699, 6.053379213, 21.0, 0.23906788,  5.219610436,
    22.0, 0.196230431, 4.484302033, 23.0, 0.160897421, 3.839943445, 24.0, 0.131795374, 3.27839108,
    25.0, 0.1078567,   2.791361596, 26.0, 0.088189063, 2.370765375, 27.0, 0.072048567, 2.008921719,
    28.0, 0.058816518, 1.698687797, 29.0, 0.047979438, 1.433525748, 30.0, 0.039111985, 1.207526336,
    31.0, 0.031862437, 1.015402928, 32.0, 0.025940415, 0.852465956, 33.0, 0.021106532, 0.714585285,
    34.0, 0.017163711, 0.598145851, 35.0, 0.013949904, 0.500000349, 36.0, 0.011332027, 0.41742159,
    37.0, 0.009200898, 0.348056286, 38.0, 0.007467036, 0.289881373, 39.0, 0.006057179, 0.241163527,
    40.0, 0.004911394, 0.200422214};

/* engine.rs:261-291  calculate_log10_dynamic_read_qual_threshold.  `base_qualities` are the ORIGINAL
 * read quals: the reference stores the modified ones under the key "HMM_BASE_QUALITIES_TAG" (:377-380)
 * but looks up "HMMQuals" (:268), so the lookup always falls back to read.qual(). */
ORACLE_API double oracle_log10_dynamic_read_qual_threshold(const uint8_t *base_qualities, size_t n,
                                                           double dynamic_read_qual_constant) {
    double sum_mean = 0.0, sum_variance = 0.0;
    for (size_t i = 0; i < n; ++i) {
        size_t bq = base_qualities[i];
        size_t entry_index = (bq <= 1) ? 0 : ((bq < 40 ? bq : 40) - 1);
        size_t mean_offset = entry_index * 3 + 1;
        sum_mean += dynamic_read_qual_thresh_lookup_table[mean_offset];
        sum_variance += dynamic_read_qual_thresh_lookup_table[mean_offset + 1];
    }
    double threshold = sum_mean + dynamic_read_qual_constant * sqrt(sum_variance);
    return threshold * -0.1;
}

/* engine.rs:293-319  log10_min_true_likelihood */
ORACLE_API double oracle_log10_min_true_likelihood(size_t qualified_read_length, double maximum_error_per_base,
                                                   int cap_likelihoods) {
    double e = ceil((double)qualified_read_length * maximum_error_per_base);
    double max_errors_for_read = cap_likelihoods ? (e < 2.0 ? e : 2.0) : e;
    return max_errors_for_read * -4.0;
}

/* engine.rs:229-239 + :244-259: the threshold function compute_read_likelihoods hands to the filter */
ORACLE_API double oracle_read_disqualification_threshold(const uint8_t *base_qualities, size_t n, int dynamic,
                                                         double read_disqualification_scale,
                                                         double expected_error_rate_per_base) {
    if (!dynamic) return oracle_log10_min_true_likelihood(n, expected_error_rate_per_base, 1);
    double dynamic_threshold = oracle_log10_dynamic_read_qual_threshold(base_qualities, n, read_disqualification_scale);
    double log10_max = oracle_log10_min_true_likelihood(n, expected_error_rate_per_base, 0);
    return dynamic_threshold < log10_max ? dynamic_threshold : log10_max;
}

/* allele_likelihoods.rs:378-444 + :457-508  normalize_likelihoods for one sample matrix `values`
 * ([allele][read], row-major, n_alleles x n_reads, as values_by_sample_index stores it).
 * reference_allele_index < 0 == None. */
ORACLE_API void oracle_normalize_likelihoods(double *values, size_t n_alleles, size_t n_reads,
                                             double maximum_likelihood_difference_cap, int symmetric,
                                             long reference_allele_index) {
    if (maximum_likelihood_difference_cap == -INFINITY) return;
    if (n_alleles == 0 || n_alleles == 1) return;
    for (size_t r = 0; r < n_reads; ++r) {
        /* search_best_allele(can_be_reference = symmetric, priorities = None) */
        int can_be_reference = symmetric;
        size_t best = (can_be_reference || reference_allele_index != 0) ? 0 : 1;
        double best_l = values[best * n_reads + r];
        for (size_t a = best + 1; a < n_alleles; ++a) {
            if (!can_be_reference && reference_allele_index == (long)a) continue;
            double cand = values[a * n_reads + r];
            if (cand > best_l) best_l = cand;
        }
        double worst_likelihood_cap = best_l + maximum_likelihood_difference_cap;
        for (size_t a = 0; a < n_alleles; ++a)
            if (values[a * n_reads + r] < worst_likelihood_cap) values[a * n_reads + r] = worst_likelihood_cap;
    }
}

/* allele_likelihoods.rs:925-1041  filter_poorly_modeled_evidence for one sample: removes reads whose
 * best likelihood over ALL alleles is below their threshold, compacts the matrix columns and fills the
 * tail with NaN.  keep[r] = 1 for surviving reads.  Returns the new evidence count. */
ORACLE_API size_t oracle_filter_poorly_modeled_evidence(double *values, size_t n_alleles, size_t n_reads,
                                                        const double *thresholds, uint8_t *keep) {
    size_t kept = 0;
    for (size_t r = 0; r < n_reads; ++r) {
        double best = -INFINITY; /* maximum_likelihood_over_all_alleles */
        for (size_t a = 0; a < n_alleles; ++a)
            if (values[a * n_reads + r] > best) best = values[a * n_reads + r];
        int removed = best < thresholds[r];
        keep[r] = (uint8_t)!removed;
        if (!removed) {
            for (size_t a = 0; a < n_alleles; ++a) values[a * n_reads + kept] = values[a * n_reads + r];
            kept += 1;
        }
    }
    if (kept < n_reads)
        for (size_t a = 0; a < n_alleles; ++a)
            for (size_t r = kept; r < n_reads; ++r) values[a * n_reads + r] = NAN;
    return kept;
}

/* ---- AlleleLikelihoods::search_best_allele (src/model/allele_likelihoods.rs:457-554) with BestAllele::new (:1142-1160),
 * as best_alleles_tie_breaking calls it (:1069-1095): can_be_reference = true, priorities present.  `values` is the
 * reference's [allele][evidence] matrix of one sample, `priorities` one i32 per allele (NULL = the `None` arm, no tie
 * breaking), `threshold` get_informative_threshold (:309-315; 0.2 for log10 likelihoods, :17).  One result per unit of
 * evidence: allele index (-1 = None: no alleles), likelihood, confidence. */
ORACLE_API void oracle_best_alleles(const double *values, size_t n_alleles, size_t n_reads, const int32_t *priorities,
                                    double threshold, int32_t *best_allele, double *likelihood, double *confidence) {
#define V(a, r) values[(a) * n_reads + (r)]
    for (size_t r = 0; r < n_reads; ++r) {
        if (n_alleles == 0) { /* :465-475 */
            best_allele[r] = -1;
            likelihood[r] = -INFINITY;
            /* BestAllele::new(-inf, -inf): (-inf) - (-inf) is NaN, NaN.abs() < EPSILON is false, so confidence = NaN */
            confidence[r] = (-INFINITY) - (-INFINITY);
            continue;
        }
        size_t best = 0, second = 0; /* :479-488, can_be_reference */
        double best_lk = V(0, r), second_lk = -INFINITY;
        for (size_t a = best + 1; a < n_alleles; ++a) { /* :490-505 */
            const double c = V(a, r);
            if (c > best_lk) {
                second = best;
                best = a;
                second_lk = best_lk;
                best_lk = c;
            } else if (c > second_lk) {
                second = a;
                second_lk = c;
            }
        }
        if (priorities && (best_lk - second_lk) < threshold) { /* :507-536 */
            int32_t best_pri = priorities[best], second_pri = priorities[second];
            for (size_t a = 0; a < n_alleles; ++a) {
                const double c = V(a, r);
                if (a == best || (best_lk - c) > threshold) continue;
                const int32_t cp = priorities[a];
                if (cp > best_pri) {
                    second = best;
                    best = a;
                    second_pri = best_pri;
                    best_pri = cp;
                } else if (cp > second_pri) {
                    second = a;
                    second_pri = cp;
                }
            }
        }
        best_lk = V(best, r); /* :538-543 */
        second_lk = second != best ? V(second, r) : -INFINITY;
        best_allele[r] = (int32_t)best;
        likelihood[r] = best_lk;
        const double d = best_lk - second_lk; /* :1149-1153 */
        confidence[r] = fabs(d) < 2.220446049250313e-16 ? 0.0 : d;
    }
#undef V
}
