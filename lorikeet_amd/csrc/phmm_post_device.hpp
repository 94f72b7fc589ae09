// Device-side pieces of the post-step shared by phmm_engine_kernels.hip (phmm_post_reads, phmm_best_alleles_kernel,
// phmm_post_best_reads) and phmm_cigar_kernels.hip (phmm_pick_reads: post-step, best allele and projection of a read in one
// launch).  Files that include this are compiled with -ffp-contract=off.
#pragma once
#include "phmm_internal.hpp"

namespace phmm {

// normalize_likelihoods + the keep / remove decision for read r (in place in p.out; small calls: also into p.out_final);
// returns keep[r]
__device__ __forceinline__ uint8_t post_read(const PostParams &p, const uint32_t r, const bool in_place_too) {
    if (r == 0 && p.status_out) *p.status_out = *p.status_in;
    const uint32_t g = p.read_region[r];
    const uint32_t nh = p.region_hap_off[g + 1] - p.region_hap_off[g];
    const uint64_t at = p.out_off[g] + (uint64_t)(r - p.region_read_off[g]) * nh;
    double *row = p.out + at;
    const int ref = p.region_ref_hap ? p.region_ref_hap[g] : -1;
    double best_all = -INFINITY;  // maximum_likelihood_over_all_alleles (:1026-1041)
    for (uint32_t a = 0; a < nh; ++a) best_all = row[a] > best_all ? row[a] : best_all;
    // normalize_likelihoods (:378-444): nothing to do for 0/1 alleles or an infinite cap
    if (nh > 1 && p.max_likelihood_difference_cap != -INFINITY) {
        // search_best_allele(can_be_reference = symmetric, priorities = None) (:457-508)
        const bool can_be_ref = p.symmetric != 0;
        uint32_t first = (can_be_ref || ref != 0) ? 0u : 1u;
        double best = row[first];
        for (uint32_t a = first + 1; a < nh; ++a) {
            if (!can_be_ref && ref == (int)a) continue;
            best = row[a] > best ? row[a] : best;
        }
        const double worst = best + p.max_likelihood_difference_cap;
        if (p.out_final) {
            for (uint32_t a = 0; a < nh; ++a) {
                const double v = row[a] < worst ? worst : row[a];
                p.out_final[at + a] = v;
                if (in_place_too) row[a] = v;  // (the best-allele search behind it reads the row from device memory)
            }
        } else {
            for (uint32_t a = 0; a < nh; ++a)
                if (row[a] < worst) row[a] = worst;
        }
        // the cap can only raise values up to `worst` <= best: the all-allele maximum is unchanged unless
        // every allele sat below `worst` (asymmetric mode with only the reference above the alts)
        best_all = best_all > worst ? best_all : worst;
    } else if (p.out_final) {
        for (uint32_t a = 0; a < nh; ++a) p.out_final[at + a] = row[a];
    }
    // filter_poorly_modeled_evidence removes evidence whose best likelihood is below its threshold (:941-958)
    const uint8_t keep = (best_all < p.threshold[r]) ? 0 : 1;
    p.keep[r] = keep;
    return keep;
}

// ---- best allele per read (AlleleLikelihoods::search_best_allele, src/model/allele_likelihoods.rs:457-554, the way
// best_alleles_tie_breaking calls it, :1069-1095: can_be_reference = true) with BestAllele::new (:1142-1160): the first
// step of realign_reads_to_their_best_haplotype (src/assembly/assembly_based_caller_utils.rs:208-246).  One thread per
// read; its row of the [read][hap] matrix is contiguous.
// best allele of read r of region g; `kept`: the evidence survived filter_poorly_modeled_evidence; `aligned`: false leaves
// the read without a reference to align to (ref_index = SW_NO_REFERENCE) although it has a best allele
__device__ __forceinline__ void best_allele_of(const BestParams &p, const uint32_t r, const uint32_t g, const bool kept, const bool aligned) {
    const uint32_t h0 = p.region_hap_off[g], nh = p.region_hap_off[g + 1] - h0;
    int32_t best_out = -1;
    double lk_out = -INFINITY, conf_out = (-INFINITY) - (-INFINITY);  // BestAllele::new(-inf, -inf): NaN (:465-475)
    if (nh && kept) {
        const double *v = p.likelihoods + p.out_off[g] + (uint64_t)(r - p.region_read_off[g]) * nh;
        uint32_t best = 0, second = 0;  // :479-488
        double best_lk = v[0], second_lk = -INFINITY;
        for (uint32_t a = 1; a < nh; ++a) {  // :490-505
            const double c = v[a];
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
        if (p.priority && (best_lk - second_lk) < p.threshold) {  // :507-536
            const int32_t *pri = p.priority + h0;
            int32_t best_pri = pri[best], second_pri = pri[second];
            for (uint32_t a = 0; a < nh; ++a) {
                const double c = v[a];
                if (a == best || (best_lk - c) > p.threshold) continue;
                const int32_t cp = pri[a];
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
        best_lk = v[best];  // :538-543
        second_lk = second != best ? v[second] : -INFINITY;
        best_out = (int32_t)best;
        lk_out = best_lk;
        const double d = best_lk - second_lk;  // :1149-1153
        conf_out = fabs(d) < 2.220446049250313e-16 ? 0.0 : d;
    }
    p.best_allele[r] = best_out;
    p.likelihood[r] = lk_out;
    p.confidence[r] = conf_out;
    if (p.ref_index) p.ref_index[r] = best_out >= 0 && aligned ? h0 + (uint32_t)best_out : SW_NO_REFERENCE;
}

// The two steps for a read with at most RH alleles, its row held in registers: ONE round trip to memory for the values
// (and one for the priorities) instead of one per pass and allele -- a thread per read is latency, not bandwidth, and a
// region per call (the reference's pattern) has only a few hundred of them.  Statement for statement post_read followed by
// best_allele_of; every loop runs over compile-time indices with `a < nh` guards.
template <int RH>
__device__ __forceinline__ void post_best_in_registers(const PostBestParams &p, const uint32_t r, const uint32_t g, const uint32_t nh) {
    const PostParams &po = p.post;
    const BestParams &bp = p.best;
    if (r == 0 && po.status_out) *po.status_out = *po.status_in;
    const uint32_t h0 = po.region_hap_off[g];
    const uint64_t at = po.out_off[g] + (uint64_t)(r - po.region_read_off[g]) * nh;
    double *row = po.out + at;
    const int ref = po.region_ref_hap ? po.region_ref_hap[g] : -1;
    double v[RH];
    int32_t pri[RH];
#pragma unroll
    for (int a = 0; a < RH; ++a) {
        v[a] = (uint32_t)a < nh ? row[a] : -INFINITY;
        pri[a] = bp.priority && (uint32_t)a < nh ? bp.priority[h0 + a] : 0;
    }
    const double threshold_r = po.threshold[r];
    // ---- post_read ----
    double best_all = -INFINITY;
#pragma unroll
    for (int a = 0; a < RH; ++a)
        if ((uint32_t)a < nh) best_all = v[a] > best_all ? v[a] : best_all;
    if (nh > 1 && po.max_likelihood_difference_cap != -INFINITY) {
        const bool can_be_ref = po.symmetric != 0;
        const uint32_t first = (can_be_ref || ref != 0) ? 0u : 1u;
        double best = first == 0 ? v[0] : v[1];
#pragma unroll
        for (int a = 1; a < RH; ++a) {
            if ((uint32_t)a < first + 1 || (uint32_t)a >= nh) continue;
            if (!can_be_ref && ref == a) continue;
            best = v[a] > best ? v[a] : best;
        }
        const double worst = best + po.max_likelihood_difference_cap;
#pragma unroll
        for (int a = 0; a < RH; ++a)
            if ((uint32_t)a < nh) {
                const bool raise = v[a] < worst;
                v[a] = raise ? worst : v[a];
                if (raise || po.out_final) row[a] = v[a];
                if (po.out_final) po.out_final[at + a] = v[a];
            }
        best_all = best_all > worst ? best_all : worst;
    } else if (po.out_final) {
#pragma unroll
        for (int a = 0; a < RH; ++a)
            if ((uint32_t)a < nh) po.out_final[at + a] = v[a];
    }
    const uint8_t keep = (best_all < threshold_r) ? 0 : 1;
    po.keep[r] = keep;
    if (p.keep_final) p.keep_final[r] = keep;
    // ---- best_allele_of ----
    int32_t best_out = -1;
    double lk_out = -INFINITY, conf_out = (-INFINITY) - (-INFINITY);
    if (nh && keep) {
        uint32_t best = 0, second = 0;
        double best_lk = v[0], second_lk = -INFINITY;
        int32_t best_pri = pri[0], second_pri = pri[0];  // the priorities of `best` / `second`, carried along
#pragma unroll
        for (int a = 1; a < RH; ++a) {
            if ((uint32_t)a >= nh) continue;
            const double c = v[a];
            if (c > best_lk) {
                second = best;
                second_lk = best_lk;
                second_pri = best_pri;
                best = a;
                best_lk = c;
                best_pri = pri[a];
            } else if (c > second_lk) {
                second = a;
                second_lk = c;
                second_pri = pri[a];
            }
        }
        double best_val = best_lk, second_val = second_lk;  // v[best], v[second] (second_val is only read when second != best)
        if (bp.priority && (best_lk - second_lk) < bp.threshold) {
            const double top = best_lk;
#pragma unroll
            for (int a = 0; a < RH; ++a) {
                if ((uint32_t)a >= nh) continue;
                const double c = v[a];
                if ((uint32_t)a == best || (top - c) > bp.threshold) continue;
                const int32_t cp = pri[a];
                if (cp > best_pri) {
                    second = best;
                    second_pri = best_pri;
                    second_val = best_val;
                    best = a;
                    best_pri = cp;
                    best_val = c;
                } else if (cp > second_pri) {
                    second = a;
                    second_pri = cp;
                    second_val = c;
                }
            }
        }
        best_lk = best_val;
        second_lk = second != best ? second_val : -INFINITY;
        best_out = (int32_t)best;
        lk_out = best_lk;
        const double d = best_lk - second_lk;
        conf_out = fabs(d) < 2.220446049250313e-16 ? 0.0 : d;
    }
    bp.best_allele[r] = best_out;
    bp.likelihood[r] = lk_out;
    bp.confidence[r] = conf_out;
    const bool aligned = !(p.skip_single_allele && nh == 1);
    if (bp.ref_index) bp.ref_index[r] = best_out >= 0 && aligned ? h0 + (uint32_t)best_out : SW_NO_REFERENCE;
}

}  // namespace phmm
