// Projection of read -> haplotype alignments onto the reference, one lane per read: everything
// AlignmentUtils::create_read_aligned_to_ref (src/reads/alignment_utils.rs:40-165) does after the Smith-Waterman call.
//   :65-72    the alignment's CIGAR through CigarBuilder (src/reads/cigar_builder.rs)
//   :84-100   Haplotype::get_consolidated_padded_cigar(1000) (src/haplotype/haplotype.rs:248-256),
//             read_start_on_reference_haplotype (:283-311), the read's start on the reference
//   :107-113  trim_cigar_by_bases (:321-386): the haplotype -> reference CIGAR from the read's start on
//   :115      apply_cigar_to_cigar (:240-281, CigarPairTransform :974-1049)
//   :116-122  left_align_indels (:425-566, normalize_alleles :585-640)
//   :126-130  the new position (left-alignment may have removed a leading deletion)
//   :135-143  append_clipped_elements_from_cigar_to_cigar (:173-213)
//   :151-161  the length check
// CIGAR elements are BAM-encoded, (length << 4) | op.  Where the reference panics or returns Err the read gets a
// negative status and is left alone.  Every lane works in its own slice of a workspace in HBM (three builders and the
// right-to-left list of left_align_indels); apply_cigar_to_cigar advances by runs instead of single bases -- the
// builder merges what the reference adds base by base, so the result is the same.
#include "phmm_cigar_device.hpp"

namespace phmm {

using namespace cigdev;

// The last kernel of a call tells the calling thread itself (ProjectParams::finish_flag): a block is one wave, its lanes'
// stores are behind the release; the block that completes the count publishes them all (release at system scope) with the flag.
__device__ __forceinline__ void count_block_in(const ProjectParams &p) {
    if (!p.finish_counter) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(p.finish_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == p.finish_target) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(p.finish_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(64) void phmm_project_kernel(const ProjectParams p) {
    const uint32_t r = p.r_begin + blockIdx.x * blockDim.x + threadIdx.x;
    // (a small launch keeps the lanes' builders in LDS: every builder operation is a dependent memory access, and a
    // region per call has too few reads to hide HBM latency behind other lanes)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_ws[];
    if (r < p.n_reads)
        project_read(p, r, p.workspace ? p.workspace + (size_t)(r - p.r_begin) * 4 * p.capacity : lds_ws + (size_t)threadIdx.x * 4 * p.capacity);
    count_block_in(p);
}

__global__ __launch_bounds__(64) void phmm_pick_reads(const PostBestParams pb, const ProjectParams p) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_ws[];
    if (r < p.n_reads) pick_read(pb, p, r, p.workspace ? p.workspace + (size_t)r * 4 * p.capacity : lds_ws + (size_t)threadIdx.x * 4 * p.capacity);
    count_block_in(p);
}

// CigarUtils::calculate_cigar (src/reads/cigar_utils.rs:358-457) behind the Smith-Waterman alignment of the padded
// sequences: the two shortcuts (:365-385), is_s_w_failure (:469-487), the padding trimmed off (:421-428), the trailing
// deletion put back for the left-alignment (:430-435), left_align_indels (:437-442), and the leading / trailing deletions
// the builder stripped restored (:444-466).  One lane per haplotype.
__global__ __launch_bounds__(64) void phmm_calculate_cigar_kernel(const CalcParams p) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.n) return;
    const uint8_t *ref_seq = p.ref_bases + p.ref_off[a] + SW_PAD_BASES, *alt_seq = p.alt_bases + p.alt_off[a] + SW_PAD_BASES;
    const uint32_t ref_len = p.ref_off[a + 1] - p.ref_off[a] - 2 * SW_PAD_BASES, alt_len = p.alt_off[a + 1] - p.alt_off[a] - 2 * SW_PAD_BASES;
    uint32_t *out = p.out_cigar + p.out_cigar_off[a];
    const uint64_t out_cap = p.out_cigar_off[a + 1] - p.out_cigar_off[a];
    uint32_t n_out = 0;
    int status = CIGAR_OK;
    auto put = [&](uint32_t e) {
        if (n_out < out_cap) out[n_out] = e;
        ++n_out;
    };
    uint32_t *ws = p.workspace + (size_t)a * 3 * p.capacity;
    Builder T, R;
    bool done = false;
    if (alt_len == 0) {  // "horrible edge case from the unit tests, where this path has no bases"
        put(elem(OP_D, ref_len));
        done = true;
    } else if (alt_len == ref_len) {  // equal lengths and at most two mismatches: all M
        uint32_t mismatches = 0;
        for (uint32_t i = 0; i < ref_len && mismatches <= 2; ++i) mismatches += alt_seq[i] != ref_seq[i];
        if (mismatches <= 2) {
            put(elem(OP_M, ref_len));
            done = true;
        }
    }
    if (!done) {
        const uint32_t *sw = p.sw_cigar + p.sw_cigar_off[a];
        const uint32_t n_sw = p.n_sw_cigar[a];
        if (p.sw_offset[a] > 0) status = CIGAR_SW_FAILURE;  // the alignment must start at the first base, given the padding
        for (uint32_t i = 0; i < n_sw && status == CIGAR_OK; ++i)
            if (op_of(sw[i]) == OP_S) status = CIGAR_SW_FAILURE;
        uint32_t lead = 0, trail = 0, lead2 = 0, trail2 = 0;
        if (status == CIGAR_OK) {
            T.init(ws, p.capacity, true);
            status = trim_by_bases(T, sw, n_sw, SW_PAD_BASES, (uint64_t)alt_len + SW_PAD_BASES - 1, &lead, &trail);
        }
        if (status == CIGAR_OK && trail > 0) status = T.n < T.cap ? (T.el[T.n++] = elem(OP_D, trail), CIGAR_OK) : CIGAR_ERR_WORKSPACE;
        if (status == CIGAR_OK)
            status = left_align(R, ws + p.capacity, p.capacity, ws + 2 * (size_t)p.capacity, T.el, T.n, ref_seq, ref_len, alt_seq, alt_len, lead,
                                &lead2, &trail2);
        if (status == CIGAR_OK) {
            if (lead + lead2 > 0) put(elem(OP_D, lead + lead2));
            for (uint32_t i = 0; i < R.n; ++i) put(R.el[i]);
            if (trail2 > 0) put(elem(OP_D, trail2));
        }
    }
    p.status[a] = status;
    p.n_out_cigar[a] = status == CIGAR_OK ? n_out : 0;
    if (status == CIGAR_OK && n_out > out_cap) *p.flags = 1u;
}

// The inputs of a small call, fetched from the pinned mirror by the compute queue itself: 16 bytes per lane straight over
// the link.  (A copy-engine transfer of a few tens of KB costs more in queueing and in the cross-engine dependency of
// the launch behind it than the transfer itself.)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) phmm_stage_in_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = __builtin_nontemporal_load(src + i);
}

hipError_t launch_stage_in(const void *host_as_device, void *dev, size_t bytes, hipStream_t stream) {
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (!n16) return hipSuccess;
    hipLaunchKernelGGL(phmm_stage_in_kernel, dim3((n16 + 255) / 256), dim3(256), 0, stream, (const u32x4 *)host_as_device, (u32x4 *)dev, n16);
    return hipGetLastError();
}

hipError_t launch_calculate_cigar(const CalcParams &p, hipStream_t stream) {
    if (!p.n) return hipSuccess;
    hipLaunchKernelGGL(phmm_calculate_cigar_kernel, dim3((p.n + 63) / 64), dim3(64), 0, stream, p);
    return hipGetLastError();
}

// (*blocks: how many blocks count themselves in -- p.finish_target comes in as the count before this launch)
hipError_t launch_project(const ProjectParams &p, hipStream_t stream, uint32_t *blocks) {
    if (blocks) *blocks = 0;
    if (p.n_reads <= p.r_begin) return hipSuccess;
    const uint32_t n = p.n_reads - p.r_begin;
    const size_t lds_per_lane = 4ull * p.capacity * 4;
    ProjectParams q = p;
    const bool in_lds = n <= 4096 && 32 * lds_per_lane <= 64 * 1024;  // workspace in LDS, half a wave per block (no attribute needed up to 64 KB)
    const uint32_t per_block = in_lds ? 32 : 64, grid = (n + per_block - 1) / per_block;
    if (in_lds) q.workspace = nullptr;
    q.finish_target = p.finish_target + grid;
    if (blocks) *blocks = grid;
    hipLaunchKernelGGL(phmm_project_kernel, dim3(grid), dim3(per_block), in_lds ? 32 * lds_per_lane : 0, stream, q);
    return hipGetLastError();
}

// (p.r_begin == 0: the launch covers the reads of the post-step)
hipError_t launch_pick(const PostBestParams &pb, const ProjectParams &p, hipStream_t stream, uint32_t *blocks) {
    if (blocks) *blocks = 0;
    if (!p.n_reads) return hipSuccess;
    if (p.r_begin != 0 || pb.post.n_reads != p.n_reads) return hipErrorInvalidValue;
    const size_t lds_per_lane = 4ull * p.capacity * 4;
    ProjectParams q = p;
    const bool in_lds = p.n_reads <= 4096 && 32 * lds_per_lane <= 64 * 1024;
    const uint32_t per_block = in_lds ? 32 : 64, grid = (p.n_reads + per_block - 1) / per_block;
    if (in_lds) q.workspace = nullptr;
    q.finish_target = p.finish_target + grid;
    if (blocks) *blocks = grid;
    hipLaunchKernelGGL(phmm_pick_reads, dim3(grid), dim3(per_block), in_lds ? 32 * lds_per_lane : 0, stream, pb, q);
    return hipGetLastError();
}

}  // namespace phmm
