// Device-side body of the pre-step (phmm_prep_reads, phmm_engine_kernels.hip; the resident region server's PREP task,
// phmm_server_kernels.hip): PairHMMLikelihoodCalculationEngine::modify_read_qualities
// (reference src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs:352-388, default branch) -- PCR indel error model
// (:502-611) + quality caps (:428-466) -- and the per-read disqualification threshold (:229-239, :244-319) from the ORIGINAL
// qualities.  Files that include this are compiled with -ffp-contract=off (the threshold must round like the reference).
#pragma once
#include "phmm_internal.hpp"

namespace phmm {

namespace prepdev {

constexpr int MAX_STR_UNIT_LENGTH = 20;  // engine.rs:98
constexpr int MAX_REPEAT_LENGTH = 100;   // engine.rs:99
constexpr uint32_t MIN_USABLE_Q = 6;     // quality_utils.rs:23

// engine.rs:23-39, (mean, variance) per base quality 1..40
static __constant__ double kDynQualTable[40][2] = {
    {5.996842844, 0.196616587}, {5.870018422, 1.388545569}, {5.401558531, 5.641990128}, {4.818940919, 10.33176216},
    {4.218758304, 14.25799688}, {3.646319832, 17.02880749}, {3.122346753, 18.64537883}, {2.654731979, 19.27521677},
    {2.244479156, 19.13584613}, {1.88893867, 18.43922003},  {1.583645342, 17.36842261}, {1.3233807, 16.07088712},
    {1.102785365, 14.65952563}, {0.916703025, 13.21718577}, {0.760361881, 11.80207947}, {0.629457387, 10.45304833},
    {0.520175654, 9.194183767}, {0.42918208, 8.038657241},  {0.353590663, 6.991779595}, {0.290923699, 6.053379213},
    {0.23906788, 5.219610436},  {0.196230431, 4.484302033}, {0.160897421, 3.839943445}, {0.131795374, 3.27839108},
    {0.1078567, 2.791361596},   {0.088189063, 2.370765375}, {0.072048567, 2.008921719}, {0.058816518, 1.698687797},
    {0.047979438, 1.433525748}, {0.039111985, 1.207526336}, {0.031862437, 1.015402928}, {0.025940415, 0.852465956},
    {0.021106532, 0.714585285}, {0.017163711, 0.598145851}, {0.013949904, 0.500000349}, {0.011332027, 0.41742159},
    {0.009200898, 0.348056286}, {0.007467036, 0.289881373}, {0.006057179, 0.241163527}, {0.004911394, 0.200422214}};

__device__ __forceinline__ bool same(const uint8_t *s, int a, int b, int len) {
    for (int i = 0; i < len; ++i)
        if (s[a + i] != s[b + i]) return false;
    return true;
}

// Number of consecutive copies of unit s[u, u+len) in s[lo, lo+tl), counted from the front (leading)
// or from the back -- VariantContextUtils::find_number_of_repetitions_main
// (src/model/variant_context_utils.rs:276-335) on sub-ranges of one string.  `first_is_unit`: the copy
// at the counted end IS the unit itself (the caller cut the unit out of that end), so its compare is
// skipped -- same count, one LDS pass less.
__device__ int repetitions(const uint8_t *s, int u, int len, int lo, int tl, bool leading, bool first_is_unit) {
    if (tl == 0) return 0;
    const int diff = tl - len;
    int n = 0;
    if (leading) {
        int start = 0;
        if (first_is_unit && diff >= 0) {
            n = 1;
            start = len;
        }
        for (; start <= diff; start += len) {
            if (!same(s, lo + start, u, len)) return n;
            ++n;
        }
    } else {
        int start = diff;
        if (first_is_unit && diff >= 0) {
            n = 1;
            start = diff - len;
        }
        for (; start >= 0; start -= len) {
            if (!same(s, lo + start, u, len)) return n;
            ++n;
        }
    }
    return n;
}

// find_tandem_repeat_units (engine.rs:528-611) -> length of the tandem repeat around `offset`.
//
// The reference tries unit lengths str = 1..20 in order and stops at the first one whose adjacent copy equals the
// unit (count > 1).  "Adjacent copy equals the unit" is a periodicity test on 2*str bases next to the offset; it is
// decided in two levels so that the common case (no repeat) costs ~40 register compares per direction instead of
// 210: a branch-free screen on the first two positions of every candidate length (bit mask per lane), then the
// exact test, from LDS, only for the ~6 % of lengths that pass it.  Positions outside the read never compare equal
// (the screen uses distinct sentinels, the exact test a range check), which is what the reference's bounds do.
// The exact, data-dependent count runs only at positions that really sit in a tandem repeat.  Same results as the
// plain loops (the GPU parity tests of the engine-level call pin this kernel to a CPU restatement of them).
__device__ int tandem_repeat_length(const uint8_t *s, int n, int offset) {
    constexpr int W = MAX_STR_UNIT_LENGTH + 2;
    uint32_t wb[W], wf[W];  // wb[d] = s[offset - d], wf[d] = s[offset + 1 + d]
#pragma unroll
    for (int d = 0; d < W; ++d) {
        wb[d] = (offset - d >= 0) ? (uint32_t)s[offset - d] : 0x100u + d;
        wf[d] = (offset + 1 + d < n) ? (uint32_t)s[offset + 1 + d] : 0x200u + d;
    }
    uint32_t cand_b = 0, cand_f = 0;  // bit str-1: unit length str passes the screen
#pragma unroll
    for (int str = 1; str <= MAX_STR_UNIT_LENGTH; ++str) {
        bool b = wb[0] == wb[str], f = wf[0] == wf[str];
        if (str >= 2) {
            b &= wb[1] == wb[1 + str];
            f &= wf[1] == wf[1 + str];
        }
        cand_b |= b ? 1u << (str - 1) : 0u;
        cand_f |= f ? 1u << (str - 1) : 0u;
    }
    // backward: unit = s[offset+1-str, offset+1), the copy before it starts at offset+1-2str (:531-560)
    int max_bw = 1, bw_u = offset, bw_len = 1;
    while (cand_b) {
        const int str = __ffs(cand_b);
        bool twice = offset + 1 - 2 * str >= 0;
        for (int d = 2; d < str && twice; ++d) twice = s[offset - d] == s[offset - d - str];
        if (twice) {
            max_bw = repetitions(s, offset + 1 - str, str, 0, offset + 1, false, true);
            bw_u = offset + 1 - str;
            bw_len = str;
            break;
        }
        cand_b &= cand_b - 1;
    }
    int max_rl = max_bw;
    if (offset < n - 1) {
        // forward: unit = s[offset+1, offset+1+str), the copy after it starts at offset+1+str (:562-587)
        int max_fw = 1, fw_len = 1;
        const int fw_u = offset + 1;
        while (cand_f) {
            const int str = __ffs(cand_f);
            bool twice = offset + 1 + 2 * str <= n;
            for (int d = 2; d < str && twice; ++d) twice = s[offset + 1 + d] == s[offset + 1 + d + str];
            if (twice) {
                max_fw = repetitions(s, offset + 1, str, offset + 1, n - offset - 1, true, true);
                fw_len = str;
                break;
            }
            cand_f &= cand_f - 1;
        }
        if (fw_len == bw_len && same(s, fw_u, bw_u, bw_len)) {
            max_rl = max_bw + max_fw;
        } else {  // the forward unit may still tile the sequence behind the offset (:589-603)
            max_bw = repetitions(s, fw_u, fw_len, 0, offset + 1, false, false);
            max_rl = max_fw + max_bw;
        }
    }
    return max_rl > MAX_REPEAT_LENGTH ? MAX_REPEAT_LENGTH : max_rl;
}

// Wave c of read r (c < p.waves_per_read: positions 64 c + lane, then strides on); `smem_wave`: 17 bytes per row of LDS of the
// wave's own, p.lds_rows rows.  Every lane of the wave arrives; r < p.n_reads.
__device__ __forceinline__ void prep_read_wave(const PrepParams &p, const uint32_t r, const uint32_t c, unsigned char *smem_wave) {
    const int lane = threadIdx.x & 63;
    const uint32_t ro = p.read_off[r];
    const int n = (int)(p.read_off[r + 1] - ro);
    if (c && (int)(64 * c) >= n) return;
    const int stride = 64 * (int)p.waves_per_read;
    // wave-private LDS: [mean f64 x rows | variance f64 x rows | bases u8 x rows]
    const uint32_t rows = p.lds_rows;
    double *s_mean = reinterpret_cast<double *>(smem_wave);
    double *s_var = s_mean + rows;
    uint8_t *s = reinterpret_cast<uint8_t *>(s_var + rows);
    for (int i = lane; i < n; i += 64) s[i] = p.read_bases[ro + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t mapq = p.mapq[r];
    if (p.dynamic_disqualification && c == 0) {  // table rows for the ORIGINAL qual (the "HMMQuals" lookup never hits, :268)
        for (int i = lane; i < n; i += 64) {
            const uint32_t q = p.base_q[ro + i];
            const uint32_t idx = q <= 1 ? 0u : min(40u, q) - 1u;
            s_mean[i] = kDynQualTable[idx][0];
            s_var[i] = kDynQualTable[idx][1];
        }
    }
    for (int i = 64 * (int)c + lane; i < n; i += stride) {
        uint32_t q = p.base_q[ro + i];
        uint32_t iq = p.ins_q ? p.ins_q[ro + i] : p.default_indel_qual;  // ReadUtils default Q45 (read_utils.rs:23)
        uint32_t dq = p.del_q ? p.del_q[ro + i] : p.default_indel_qual;
        if (p.pcr_cache && i < n - 1) {  // apply_pcr_error_model touches every base but the last (:513-523)
            const uint32_t c = p.pcr_cache[tandem_repeat_length(s, n, i)];
            iq = min(iq, c);
            dq = min(dq, c);
        }
        // cap_minimum_read_qualities (:436-457)
        if (!p.disable_cap_to_mapq) q = min(q, mapq);
        if (q < p.base_quality_score_threshold) q = MIN_USABLE_Q;
        if (iq < MIN_USABLE_Q) iq = MIN_USABLE_Q;
        if (dq < MIN_USABLE_Q) dq = MIN_USABLE_Q;
        p.out_q[ro + i] = (uint8_t)q;
        p.out_ins[ro + i] = (uint8_t)iq;
        p.out_del[ro + i] = (uint8_t)dq;
        p.out_gcp[ro + i] = p.constant_gcp;  // PairHMMInputScoreImputator::gap_continuation_penalties (:649-651)
    }
    // Threshold handed to filter_poorly_modeled_evidence (:229-239).  The per-base table values were
    // gathered in parallel above; one lane adds them from LDS in read order, which keeps the reference's
    // summation order (bit-identical threshold) at ~1 us per read.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0 && c == 0) {
        const double e = ceil((double)n * p.expected_error_rate_per_base);  // log10_min_true_likelihood (:293-319)
        double thr;
        if (!p.dynamic_disqualification) {
            thr = fmin(2.0, e) * -4.0;
        } else {
            double sum_mean = 0.0, sum_var = 0.0;  // calculate_log10_dynamic_read_qual_threshold (:261-291)
            for (int i = 0; i < n; ++i) {
                sum_mean += s_mean[i];
                sum_var += s_var[i];
            }
            const double dyn = (sum_mean + p.read_disqualification_scale * sqrt(sum_var)) * -0.1;
            const double cap = e * -4.0;
            thr = dyn < cap ? dyn : cap;  // dynamic_log10_min_likelihood_model (:244-259)
        }
        p.threshold[r] = thr;
    }
}

}  // namespace prepdev

}  // namespace phmm
