// The recurrence in the reference's own operation order, one thread per (read, haplotype) pair.
//
// Everything the fast kernels do differently from the scalar arm of the reference -- row constants with three factors
// folded in, fused multiply-adds, the chained kernel's common start value -- is an exact rescaling or a rounding in
// the last place as long as the numbers are normal f64.  Close to the bottom of the f64 range that stops being true:
// below log10 L ~ -617 the scaled row sum enters the denormal range and the result (including the point where it
// becomes -inf) depends on every single rounding.  This file evaluates
//     M[i][j] = prior * ((M[i-1][j-1]*mm + I[i-1][j-1]*im) + D[i-1][j-1]*im)
//     I[i][j] = M[i-1][j]*mi + I[i-1][j]*ii
//     D[i][j] = M[i][j-1]*md + D[i][j-1]*dd          (reference src/pair_hmm/pair_hmm.rs:573-593)
//     result  = log10(((..(0 + (M[R][1]+I[R][1])) + ..) + (M[R][H]+I[R][H]))) - log10(2^1020)   (:598-614)
// with D[0][j] = 2^1020 / H (:515-529), one rounding per multiply and per add (no contraction: __dmul_rn / __dadd_rn,
// and the file is built with -ffp-contract=off), priors and transitions from the same tables the reference's formulas
// give (bit-identical to the oracle's, tests/test_abi.py).  Two users:
//   * phmm_rescue: after the fast kernels, every pair whose result is below kRescueBelow (-600), -inf or NaN is
//     recomputed here, so the underflow band agrees with the reference to the last place (tests/test_underflow_band.py);
//   * phmm_forward_generic: shapes outside the register kernels (haplotype > 2048 columns, reads beyond the LDS
//     staging): a correctness net, not tuned.
// Rolling rows live in global scratch, interleaved by thread so that neighbouring threads touch neighbouring addresses.
#include "phmm_internal.hpp"

namespace phmm {

namespace {

struct Rows {
    double *S;
    uint64_t W, nthreads, tid;
    __device__ __forceinline__ double &at(int arr, uint64_t j) const { return S[((uint64_t)arr * W + j) * nthreads + tid]; }
};

__device__ double exact_pair(const ForwardParams &p, const uint32_t ro, const int R, const uint32_t ho, const int H,
                             const Rows &rw) {
    const double c = p.initial_condition / (double)H;  // :515-517
    int prv = 0, cur = 3;
    for (int j = 0; j <= H; ++j) {
        rw.at(prv + 0, j) = 0.0;
        rw.at(prv + 1, j) = 0.0;
        rw.at(prv + 2, j) = c;
    }
    for (int i = 0; i < R; ++i) {
        const uint32_t x = p.read_bases[ro + i], q = p.base_q[ro + i], iq = p.ins_q[ro + i], dq = p.del_q[ro + i],
                       g = p.gcp[ro + i];
        const uint32_t mx = max(iq, dq), mn = min(iq, dq);
        // pair_hmm_model.rs:142-156: [mm, mi, md, im, ii, dd]
        const double mm = p.mm[((mx * (mx + 1)) >> 1) + mn], mi = p.eps[iq], md = p.eps[dq], ii = p.eps[g];
        const double im = __dsub_rn(1.0, ii);
        const double pm = __dsub_rn(1.0, p.eps[q]);               // qual_to_prob
        const double px = (x == 'N') ? pm : p.eps_mis[q];         // pair_hmm.rs:643-651
        double dM = rw.at(prv + 0, 0), dI = rw.at(prv + 1, 0), dD = rw.at(prv + 2, 0);
        double leftM = 0.0, leftD = 0.0;  // column 0 of rows >= 1 is never written by the reference: zeros
        rw.at(cur + 0, 0) = 0.0;
        rw.at(cur + 1, 0) = 0.0;
        rw.at(cur + 2, 0) = 0.0;
        for (int j = 1; j <= H; ++j) {
            const uint32_t y = p.hap_bases[ho + j - 1];
            const double uM = rw.at(prv + 0, j), uI = rw.at(prv + 1, j), uD = rw.at(prv + 2, j);
            const double prior = (x == y || y == 'N') ? pm : px;
            const double Mn = __dmul_rn(prior, __dadd_rn(__dadd_rn(__dmul_rn(dM, mm), __dmul_rn(dI, im)), __dmul_rn(dD, im)));
            const double In = __dadd_rn(__dmul_rn(uM, mi), __dmul_rn(uI, ii));
            const double Dn = __dadd_rn(__dmul_rn(leftM, md), __dmul_rn(leftD, ii));
            rw.at(cur + 0, j) = Mn;
            rw.at(cur + 1, j) = In;
            rw.at(cur + 2, j) = Dn;
            dM = uM;
            dI = uI;
            dD = uD;
            leftM = Mn;
            leftD = Dn;
        }
        const int tmp = prv;
        prv = cur;
        cur = tmp;
    }
    double s = 0.0;
    for (int j = 1; j <= H; ++j) s = __dadd_rn(s, __dadd_rn(rw.at(prv + 0, j), rw.at(prv + 1, j)));
    return log10(s) - p.initial_condition_log10;
}

}  // namespace

__global__ __launch_bounds__(256) void phmm_forward_generic(const GenericParams gp) {
    const ForwardParams &p = gp.f;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Rows rw{gp.scratch, (uint64_t)gp.max_h + 1, nthreads, tid};
    for (uint64_t pair = tid; pair < gp.n_pairs; pair += nthreads) {
        // item = last i with pair_first[i] <= pair
        uint32_t lo = 0, hi = p.n_items;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (gp.pair_first[mid] <= pair) lo = mid; else hi = mid;
        }
        const uint32_t r = p.class_reads ? p.class_reads[lo] : lo;
        const uint32_t a = (uint32_t)(pair - gp.pair_first[lo]);
        const uint32_t reg = p.read_region[r];
        const uint32_t ro = p.read_off[r];
        const int R = (int)(p.read_off[r + 1] - ro);
        const uint32_t h0 = p.region_hap_off[reg];
        const int Nh = (int)(p.region_hap_off[reg + 1] - h0);
        const uint32_t ho = p.hap_off[h0 + a];
        const int H = (int)(p.hap_off[h0 + a + 1] - ho);
        const double v = exact_pair(p, ro, R, ho, H, rw);
        p.out[p.out_off[reg] + (uint64_t)(r - p.region_read_off[reg]) * (uint64_t)Nh + a] = v;
        if (!(v <= 0.0)) atomicOr(p.status, STATUS_POSITIVE);
    }
}

// One thread per read: look at the read's row of results, redo what lies below kRescueBelow.
__global__ __launch_bounds__(64) void phmm_rescue(const RescueParams rp) {
    const ForwardParams &p = rp.f;
    if (!rp.force && !(__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & STATUS_RESCUE)) return;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Rows rw{rp.scratch, (uint64_t)rp.max_h + 1, nthreads, tid};
    for (uint64_t r = tid; r < rp.n_reads; r += nthreads) {
        const uint32_t reg = p.read_region[r];
        const uint32_t ro = p.read_off[r];
        const int R = (int)(p.read_off[r + 1] - ro);
        const uint32_t h0 = p.region_hap_off[reg];
        const int Nh = (int)(p.region_hap_off[reg + 1] - h0);
        double *row = p.out + p.out_off[reg] + (uint64_t)((uint32_t)r - p.region_read_off[reg]) * (uint64_t)Nh;
        for (int a = 0; a < Nh; ++a) {
            if (row[a] >= kRescueBelow) {  // false for NaN too
                if (row[a] > 0.0) atomicOr(p.status, STATUS_POSITIVE_FINAL);
                continue;
            }
            const uint32_t ho = p.hap_off[h0 + a];
            const int H = (int)(p.hap_off[h0 + a + 1] - ho);
            const double v = exact_pair(p, ro, R, ho, H, rw);
            row[a] = v;
            if (!(v <= 0.0)) atomicOr(p.status, STATUS_POSITIVE | STATUS_POSITIVE_FINAL);
        }
    }
}

hipError_t launch_generic(const GenericParams &gp, hipStream_t stream) {
    // scratch was sized for exactly this grid by the planner
    hipLaunchKernelGGL(phmm_forward_generic, dim3(gp.n_blocks), dim3(256), 0, stream, gp);
    return hipGetLastError();
}

hipError_t launch_rescue(const RescueParams &rp, hipStream_t stream) {
    if (!rp.n_reads) return hipSuccess;
    hipLaunchKernelGGL(phmm_rescue, dim3(rp.n_blocks), dim3(64), 0, stream, rp);
    return hipGetLastError();
}

}  // namespace phmm
