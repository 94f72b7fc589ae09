// phmm_sw_align (include/phmm.h): host side of the Smith-Waterman aligner -- staging, worker geometry, status.
// The alignment itself (matrix, backtrack, CIGAR) runs in phmm_sw_kernels.hip; there is no CPU path here.
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <new>
#include <string>

#include "phmm_host.hpp"
#include "phmm_internal.hpp"

using namespace phmm;

namespace {

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct DevGuard {
    int prev = -1, dev;
    explicit DevGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DevGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

bool ok(phmm_handle *h, hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    h->err = std::string(what) + ": " + hipGetErrorString(e);
    h->err_code = PHMM_ERR_HIP;
    return false;
}

}  // namespace

extern "C" int phmm_sw_align(phmm_handle *h, uint32_t n_alignments, const uint32_t *ref_off, const uint8_t *ref_bases,
                             const uint32_t *alt_off, const uint8_t *alt_bases, const phmm_sw_parameters *params,
                             int overhang_strategy, const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar,
                             int32_t *alignment_offset) {
    if (!h) return PHMM_ERR_INVALID_ARG;
    try {
        h->err_code = PHMM_OK;
        auto fail = [&](const char *msg) {
            h->err = msg;
            return h->err_code = PHMM_ERR_INVALID_ARG;
        };
        if (!params) return fail("phmm_sw_align: null parameters");
        if (overhang_strategy < PHMM_SW_SOFTCLIP || overhang_strategy > PHMM_SW_IGNORE)
            return fail("phmm_sw_align: unknown overhang strategy");
        if (!n_alignments) return PHMM_OK;
        if (!ref_off || !alt_off || !cigar_off || !n_cigar || !alignment_offset) return fail("phmm_sw_align: null array");
        if (ref_off[0] != 0 || alt_off[0] != 0 || cigar_off[0] != 0) return fail("phmm_sw_align: offset arrays must start at 0");
        uint32_t max_ref = 0, max_alt = 0;
        for (uint32_t a = 0; a < n_alignments; ++a) {
            if (ref_off[a + 1] < ref_off[a] || alt_off[a + 1] < alt_off[a] || cigar_off[a + 1] < cigar_off[a])
                return fail("phmm_sw_align: offsets not monotonic");
            // the reference asserts / panics on empty input (smith_waterman_aligner.rs:65-68, :132-134)
            if (ref_off[a + 1] == ref_off[a] || alt_off[a + 1] == alt_off[a])
                return fail("phmm_sw_align: non-empty sequences are required for the Smith-Waterman calculation");
            max_ref = std::max(max_ref, ref_off[a + 1] - ref_off[a]);
            max_alt = std::max(max_alt, alt_off[a + 1] - alt_off[a]);
        }
        if (max_ref > 32000 || max_alt > 32000)  // gap lengths travel as int16 in the backtrack matrix
            return fail("phmm_sw_align: sequences longer than 32 000 bases are not supported");
        const size_t rb = ref_off[n_alignments], ab = alt_off[n_alignments];
        const uint64_t n_cig = cigar_off[n_alignments];
        if (!ref_bases || !alt_bases || (n_cig && !cigar)) return fail("phmm_sw_align: null array");

        DevGuard dg(h->device);
        // ---- geometry ---------------------------------------------------------------------------------------------
        // K columns per lane so that one strip of 16 x K columns covers the longest alternate sequence (up to 512
        // columns; longer ones take several strips of 512)
        int K = kSwK[kNumSwK - 1];
        for (int i = kNumSwK - 1; i >= 0; --i)
            if ((size_t)kSwK[i] * 16 >= max_alt) K = kSwK[i];
        const size_t strips = (max_alt + 16ull * K - 1) / (16ull * K);
        const size_t lds_ref = (max_ref + 15) / 16 * 16, lds_alt = (max_alt + 15) / 16 * 16;
        // per alignment: the two sequences, the bottom row, and (several strips only) the strip edge, three i32 per row
        const size_t lds_group = (lds_ref + lds_alt + 4ull * (max_alt + 1) + (strips > 1 ? 12ull * (max_ref + 1) : 0) + 15) / 16 * 16;
        // four alignments share a wave; sequences so long that four do not fit a block's LDS get the wave to themselves
        const size_t gpb = 4 * lds_group <= 160 * 1024 ? 4 : 1;
        const size_t lds = gpb * lds_group;
        if (lds > 160 * 1024) return fail("phmm_sw_align: sequences too long for the LDS staging (about 8 000 bases each)");
        // int16 entries per block: strips x (rows + 15) steps x 64 lanes x K columns, whether a block holds four alignments or one
        // blocks (one wave, four alignments each): what LDS and registers let a CU hold (at most 24 waves: the kernel is
        // latency-bound), capped by the work and by 6 GB of backtrack storage
        static const size_t max_per_cu = getenv("PHMM_SW_WAVES_PER_CU") ? (size_t)atoi(getenv("PHMM_SW_WAVES_PER_CU")) : 24;
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>(max_per_cu, (160 * 1024) / lds));
        size_t workers = std::min<size_t>(256 * per_cu, ((size_t)n_alignments + gpb - 1) / gpb);
        const size_t slab_stride = strips * (size_t)(max_ref + 16) * 64 * K;
        workers = std::max<size_t>(1, std::min<size_t>(workers, (6ull << 30) / (slab_stride * 2)));
        const size_t slab_bytes = workers * slab_stride * 2;
        phmm_handle::SwWork &W = h->swork;
        hipStream_t S = h->streams[0];
        if (W.slab_bytes < slab_bytes) {
            (void)hipStreamSynchronize(S);
            if (W.slab) (void)hipFree(W.slab);
            W.slab = nullptr;
            W.slab_bytes = 0;
            if (!ok(h, hipMalloc((void **)&W.slab, slab_bytes), "hipMalloc(sw backtrack)")) return PHMM_ERR_HIP;
            W.slab_bytes = slab_bytes;
        }
        // ---- staging: [ref_off | alt_off | cigar_off | ref | alt] in, [counter+status | n_cigar | offsets | cigar] out
        const size_t o_ro = 0, o_ao = o_ro + up256(4ull * (n_alignments + 1)), o_co = o_ao + up256(4ull * (n_alignments + 1)),
                     o_rb = o_co + up256(8ull * (n_alignments + 1)), o_ab = o_rb + up256(rb), in_bytes = o_ab + up256(ab);
        const size_t o_st = in_bytes, o_nc = o_st + 256, o_of = o_nc + up256(4ull * n_alignments),
                     o_cg = o_of + up256(4ull * n_alignments), total = o_cg + up256(4ull * n_cig);
        if (W.cap < total) {
            (void)hipStreamSynchronize(S);
            if (W.dev) (void)hipFree(W.dev);
            if (W.host) (void)hipHostFree(W.host);
            W.dev = W.host = nullptr;
            W.cap = 0;
            const size_t cap = std::max<size_t>(total + total / 2, 1 << 20);
            if (!ok(h, hipMalloc((void **)&W.dev, cap), "hipMalloc(sw staging)") ||
                !ok(h, hipHostMalloc((void **)&W.host, cap, hipHostMallocDefault), "hipHostMalloc(sw staging)"))
                return PHMM_ERR_HIP;
            W.cap = cap;
        }
        memcpy(W.host + o_ro, ref_off, 4ull * (n_alignments + 1));
        memcpy(W.host + o_ao, alt_off, 4ull * (n_alignments + 1));
        memcpy(W.host + o_co, cigar_off, 8ull * (n_alignments + 1));
        memcpy(W.host + o_rb, ref_bases, rb);
        memcpy(W.host + o_ab, alt_bases, ab);
        memset(W.host + o_st, 0, 256);
        h->stat_staged_bytes += rb + ab;
        if (!ok(h, hipMemcpyAsync(W.dev, W.host, in_bytes + 256, hipMemcpyHostToDevice, S), "H2D sw")) return PHMM_ERR_HIP;
        SwParams p{};
        p.n_alignments = n_alignments;
        p.ref_off = (const uint32_t *)(W.dev + o_ro);
        p.alt_off = (const uint32_t *)(W.dev + o_ao);
        p.cigar_off = (const uint64_t *)(W.dev + o_co);
        p.ref_bases = (const uint8_t *)(W.dev + o_rb);
        p.alt_bases = (const uint8_t *)(W.dev + o_ab);
        p.w_match = params->match_value;
        p.w_mismatch = params->mismatch_penalty;
        p.w_open = params->gap_open_penalty;
        p.w_extend = params->gap_extend_penalty;
        p.strategy = overhang_strategy;
        p.cigar = (uint32_t *)(W.dev + o_cg);
        p.n_cigar = (uint32_t *)(W.dev + o_nc);
        p.alignment_offset = (int32_t *)(W.dev + o_of);
        p.slab = W.slab;
        p.slab_stride = slab_stride;
        p.status = (uint32_t *)(W.dev + o_st + 64);
        p.max_ref = max_ref;
        p.max_alt = max_alt;
        p.lds_ref_bytes = (uint32_t)lds_ref;
        p.lds_alt_bytes = (uint32_t)lds_alt;
        p.lds_group_bytes = (uint32_t)lds_group;
        p.groups_per_block = (uint32_t)gpb;
        if (!W.ev0 && (!ok(h, hipEventCreate(&W.ev0), "hipEventCreate") || !ok(h, hipEventCreate(&W.ev1), "hipEventCreate")))
            return PHMM_ERR_HIP;
        (void)hipEventRecord(W.ev0, S);
        const bool launched = ok(h, launch_sw(K, p, (uint32_t)workers, lds, S), "phmm_sw_align_kernel");
        (void)hipEventRecord(W.ev1, S);
        if (!launched ||
            !ok(h, hipMemcpyAsync(W.host + o_st, W.dev + o_st, total - o_st, hipMemcpyDeviceToHost, S), "D2H sw") ||
            !ok(h, hipStreamSynchronize(S), "sync(sw)"))
            return PHMM_ERR_HIP;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, W.ev0, W.ev1) == hipSuccess) W.last_kernel_us = (uint64_t)(ms * 1e3f);
        W.last_backtrack_bytes = 0;
        for (uint32_t a = 0; a < n_alignments; ++a)  // what the kernel stores per alignment: (rows + 15) steps x 16 K columns x 2 B per strip
            W.last_backtrack_bytes += (uint64_t)((alt_off[a + 1] - alt_off[a] + 16ull * K - 1) / (16ull * K)) *
                                      (ref_off[a + 1] - ref_off[a] + 15ull) * 16ull * K * 2ull;
        memcpy(n_cigar, W.host + o_nc, 4ull * n_alignments);
        memcpy(alignment_offset, W.host + o_of, 4ull * n_alignments);
        if (n_cig) memcpy(cigar, W.host + o_cg, 4ull * n_cig);
        const uint32_t st = *(const uint32_t *)(W.host + o_st + 64);
        if (st & SW_STATUS_CAPACITY) {
            h->err = "phmm_sw_align: a CIGAR needs more elements than its slot holds (n_cigar has the sizes)";
            return h->err_code = PHMM_ERR_CIGAR_CAPACITY;
        }
        return PHMM_OK;
    } catch (const std::bad_alloc &) {
        h->err = "phmm_sw_align: out of host memory";
        return h->err_code = PHMM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        h->err = std::string("phmm_sw_align: ") + e.what();
        return h->err_code = PHMM_ERR_INTERNAL;
    }
}
