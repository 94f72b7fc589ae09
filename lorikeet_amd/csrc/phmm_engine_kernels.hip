// gfx950 kernels of the engine-level steps around the PairHMM forward kernel:
//
//   phmm_prep_reads   PairHMMLikelihoodCalculationEngine::modify_read_qualities
//                     (reference src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs:352-388, default
//                     branch): PCR indel error model (:502-611) + quality caps (:428-466), and the
//                     per-read disqualification threshold (:229-239, :244-319) from the ORIGINAL quals.
//   phmm_post_reads   AlleleLikelihoods::normalize_likelihoods (src/model/allele_likelihoods.rs:378-508)
//                     and the keep / remove decision of filter_poorly_modeled_evidence (:925-1041).
//
// Both are byte / small-integer work over O(sum R) and O(Nr*Nh) data: one wave per read for the pre-step
// (read staged in LDS, one lane per base position), one thread per read for the post-step.
#include "phmm_internal.hpp"
#include "phmm_post_device.hpp"
#include "phmm_prep_device.hpp"

namespace phmm {

typedef uint32_t prep_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void phmm_prep_reads(const PrepParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t prep_blocks = gridDim.x - (p.stage_n16 + 255) / 256;
    if (blockIdx.x >= prep_blocks) {  // the copy of the staged inputs, side by side with the pre-step
        const uint32_t i = (blockIdx.x - prep_blocks) * 256u + threadIdx.x;
        if (i < p.stage_n16)
            reinterpret_cast<prep_u32x4 *>(p.stage_dst)[i] = __builtin_nontemporal_load(reinterpret_cast<const prep_u32x4 *>(p.stage_src) + i);
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a small batch spreads every read over several waves (a region per call: 128 reads would occupy 128 of the chip's
    // 1024 SIMDs for three rounds of 64 positions each; with one round per wave the call is 12 us shorter)
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + wave;
    const uint32_t r = gw / p.waves_per_read, c = gw % p.waves_per_read;
    if (r >= p.n_reads) return;
    prepdev::prep_read_wave(p, r, c, smem + (size_t)wave * p.lds_rows * 17);
}

__global__ __launch_bounds__(256) void phmm_post_reads(const PostParams p) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n_reads) return;
    (void)post_read(p, r, false);
}

__global__ __launch_bounds__(256) void phmm_best_alleles_kernel(const BestParams p) {
    const uint32_t r = p.r_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n_reads) return;
    uint32_t lo = 0, hi = p.n_regions;  // the region of read r: the last g with region_read_off[g] <= r
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (p.region_read_off[mid] <= r) lo = mid;
        else hi = mid;
    }
    best_allele_of(p, r, lo, !p.keep || p.keep[r], true);
}

// Both steps for the same read in one launch (phmm_region_compute): the row the post-step has just normalised is the row
// the best-allele search reads -- nothing leaves the device in between.
__global__ __launch_bounds__(256) void phmm_post_best_reads(const PostBestParams p) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.post.n_reads) return;
    const uint32_t g = p.post.read_region[r];
    const uint32_t nh = p.post.region_hap_off[g + 1] - p.post.region_hap_off[g];
    if (nh <= 16) {
        post_best_in_registers<16>(p, r, g, nh);
        return;
    }
    const uint8_t keep = post_read(p.post, r, true);
    if (p.keep_final) p.keep_final[r] = keep;
    best_allele_of(p.best, r, g, keep != 0, !(p.skip_single_allele && nh == 1));
}

hipError_t launch_post_best(const PostBestParams &p, hipStream_t stream) {
    if (!p.post.n_reads) return hipSuccess;
    hipLaunchKernelGGL(phmm_post_best_reads, dim3((p.post.n_reads + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_best_alleles(const BestParams &p, hipStream_t stream) {
    if (p.n_reads <= p.r_begin) return hipSuccess;
    hipLaunchKernelGGL(phmm_best_alleles_kernel, dim3((p.n_reads - p.r_begin + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_prep(const PrepParams &p, hipStream_t stream) {
    if (!p.n_reads) return hipSuccess;
    const int wpb = 4;
    const size_t lds = (size_t)p.lds_rows * 17 * wpb;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(phmm_prep_reads),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const size_t waves = (size_t)p.n_reads * p.waves_per_read;
    hipLaunchKernelGGL(phmm_prep_reads, dim3((unsigned)((waves + wpb - 1) / wpb + (p.stage_n16 + 255) / 256)), dim3(64 * wpb), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_post(const PostParams &p, hipStream_t stream) {
    if (!p.n_reads) return hipSuccess;
    hipLaunchKernelGGL(phmm_post_reads, dim3((p.n_reads + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace phmm
