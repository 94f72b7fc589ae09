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
        {
            // the kernel carries scores times four with a two-bit tag and leaves out the reference's clamp at -1e8
            // (MATRIX_MIN_CUTOFF): both are exact as long as no score can get near that clamp
            const int64_t big = std::max(std::max(std::llabs((long long)params->match_value), std::llabs((long long)params->mismatch_penalty)),
                                         std::max(std::llabs((long long)params->gap_open_penalty), std::llabs((long long)params->gap_extend_penalty)));
            if (big * ((int64_t)max_ref + max_alt + 2) >= 100000000)
                return fail("phmm_sw_align: parameters too large for these sequence lengths (|weight| x (ref + alt) must stay below 1e8)");
        }
        const size_t rb = ref_off[n_alignments], ab = alt_off[n_alignments];
        const uint64_t n_cig = cigar_off[n_alignments];
        if (!ref_bases || !alt_bases || (n_cig && !cigar)) return fail("phmm_sw_align: null array");

        DevGuard dg(h->device);
        // ---- geometry ---------------------------------------------------------------------------------------------
        // L lanes per alignment, K columns per lane, so that one strip of L x K columns covers the longest alternate
        // sequence: eight lanes (eight alignments per wave, fewer steps lost to the skew and less per-step work per
        // cell) while 8 x 20 columns suffice, sixteen beyond; more than 512 columns take several strips of 512
        int L = 16, K = kSwK16[kNumSwK16 - 1];
        const int force_L = h->sw.sw_lanes;
        if (force_L != 16 && (max_alt <= 8 * 20 || force_L == 8)) {
            L = 8;
            K = kSwK8[kNumSwK8 - 1];
            for (int i = kNumSwK8 - 1; i >= 0; --i)
                if ((size_t)kSwK8[i] * 8 >= max_alt) K = kSwK8[i];
        } else {
            for (int i = kNumSwK16 - 1; i >= 0; --i)
                if ((size_t)kSwK16[i] * 16 >= max_alt) K = kSwK16[i];
        }
        const size_t strip_cols = (size_t)L * K;
        const size_t strips = (max_alt + strip_cols - 1) / strip_cols;
        const size_t lds_ref = (max_ref + 15) / 16 * 16, lds_alt = (max_alt + 15) / 16 * 16;
        // per alignment: the two sequences, the bottom row, and (several strips only) the strip edge, two i32 per row
        const size_t lds_group = (lds_ref + lds_alt + 4ull * (max_alt + 1) + (strips > 1 ? 8ull * (max_ref + 1) : 0) + 15) / 16 * 16;
        // 64 / L alignments share a wave; sequences so long that they do not fit a block's LDS together get the wave to themselves
        const size_t gpb = (64 / L) * lds_group <= 160 * 1024 ? 64 / L : 1;
        const size_t lds = gpb * lds_group;
        if (lds > 160 * 1024) return fail("phmm_sw_align: sequences too long for the LDS staging (about 8 000 bases each)");
        // persistent blocks (one wave each, `gpb` alignments at a time): exactly what the chip holds at once -- more would
        // queue behind the first ones and leave the last round ragged -- capped by the work and by 6 GB of backtrack storage
        int per_cu = sw_blocks_per_cu(L, K, lds);
        if (per_cu <= 0) {
            h->err = "phmm_sw_align: the kernel does not fit a compute unit";
            return h->err_code = PHMM_ERR_INTERNAL;
        }
        if (h->sw.sw_waves_per_cu > 0) per_cu = std::min(per_cu, h->sw.sw_waves_per_cu);
        // backtrack flags per block: strips x (rows + L - 1) steps x 2 ceil(K / 16) dwords x 64 lanes (four bits per cell)
        const size_t flag_words = 2 * (((size_t)K + 15) / 16);
        const size_t slab_stride = strips * (size_t)(max_ref + L) * flag_words * 64;
        const size_t max_workers = std::max<size_t>(1, std::min<size_t>(256 * (size_t)per_cu, (6ull << 30) / (slab_stride * 4)));
        // pieces: the bases of piece c+1 are staged and copied while piece c computes (the kernels follow each other on
        // one stream and share the slabs).  A piece is a whole number of rounds of the persistent blocks, so that only
        // the last piece of a call ends on a partly filled round.
        const size_t tasks = ((size_t)n_alignments + gpb - 1) / gpb, rounds = (tasks + max_workers - 1) / max_workers;
        int n_chunks = 1;
        if (h->sw.sw_chunks > 0)
            n_chunks = h->sw.sw_chunks;
        else if (rb + ab >= (8u << 20))
            n_chunks = (int)std::min<size_t>(phmm_handle::SwWork::kMaxChunks, rounds);
        n_chunks = std::max(1, std::min<int>({n_chunks, phmm_handle::SwWork::kMaxChunks, (int)n_alignments}));
        uint32_t cut[phmm_handle::SwWork::kMaxChunks + 1];
        cut[0] = 0;
        if (h->sw.sw_chunks > 0) {
            for (int c = 1; c < n_chunks; ++c) cut[c] = (uint32_t)((uint64_t)n_alignments * c / n_chunks);  // forced: equal shares, ragged
        } else {
            const size_t rounds_per_chunk = (rounds + n_chunks - 1) / n_chunks;
            for (int c = 1; c < n_chunks; ++c)
                cut[c] = (uint32_t)std::min<uint64_t>(n_alignments, (uint64_t)c * rounds_per_chunk * max_workers * gpb);
        }
        cut[n_chunks] = n_alignments;
        size_t most = 0;
        for (int c = 0; c < n_chunks; ++c) most = std::max<size_t>(most, cut[c + 1] - cut[c]);
        const size_t slab_bytes = std::min<size_t>(max_workers, (most + gpb - 1) / gpb) * slab_stride * 4;
        phmm_handle::SwWork &W = h->swork;
        hipStream_t S = h->streams[0], S_in = h->streams[1];
        if (W.slab_bytes < slab_bytes) {
            (void)hipStreamSynchronize(S);
            if (W.slab) (void)hipFree(W.slab);
            W.slab = nullptr;
            W.slab_bytes = 0;
            if (!ok(h, hipMalloc((void **)&W.slab, slab_bytes), "hipMalloc(sw backtrack)")) return PHMM_ERR_HIP;
            W.slab_bytes = slab_bytes;
        }
        // ---- staging: [status | ref_off | alt_off | cigar_off | ref | alt] in, [status | n_cigar | offsets | cigar] out
        const size_t o_ro = 256, o_ao = o_ro + up256(4ull * (n_alignments + 1)), o_co = o_ao + up256(4ull * (n_alignments + 1)),
                     o_rb = o_co + up256(8ull * (n_alignments + 1)), o_ab = o_rb + up256(rb), in_bytes = o_ab + up256(ab);
        const size_t o_st = in_bytes, o_nc = o_st + 256, o_of = o_nc + up256(4ull * n_alignments),
                     o_cg = o_of + up256(4ull * n_alignments), total = o_cg + up256(4ull * n_cig);
        if (W.cap < total) {
            (void)hipStreamSynchronize(S);
            (void)hipStreamSynchronize(S_in);
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
        for (int c = 0; c < n_chunks; ++c)
            if (!W.ev_in[c] && (!ok(h, hipEventCreateWithFlags(&W.ev_in[c], hipEventDisableTiming), "hipEventCreate") ||
                                !ok(h, hipEventCreateWithFlags(&W.ev_out[c], hipEventDisableTiming), "hipEventCreate") ||
                                !ok(h, hipEventCreate(&W.ev_k0[c]), "hipEventCreate") || !ok(h, hipEventCreate(&W.ev_k1[c]), "hipEventCreate")))
                return PHMM_ERR_HIP;
        SwParams p{};
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
        p.status = (uint32_t *)(W.dev + 64);
        p.max_ref = max_ref;
        p.max_alt = max_alt;
        p.lds_ref_bytes = (uint32_t)lds_ref;
        p.lds_alt_bytes = (uint32_t)lds_alt;
        p.lds_group_bytes = (uint32_t)lds_group;
        p.groups_per_block = (uint32_t)gpb;
        // the offset arrays and the status word travel with the first piece
        memset(W.host, 0, 256);
        memcpy(W.host + o_ro, ref_off, 4ull * (n_alignments + 1));
        memcpy(W.host + o_ao, alt_off, 4ull * (n_alignments + 1));
        memcpy(W.host + o_co, cigar_off, 8ull * (n_alignments + 1));
        if (!ok(h, hipMemcpyAsync(W.dev, W.host, o_rb, hipMemcpyHostToDevice, S_in), "H2D sw")) return PHMM_ERR_HIP;
        h->stat_staged_bytes += rb + ab;
        bool good = true;
        for (int c = 0; c < n_chunks && good; ++c) {
            const uint32_t a0 = cut[c], a1 = cut[c + 1];
            const size_t r0 = ref_off[a0], r1 = ref_off[a1], q0 = alt_off[a0], q1 = alt_off[a1];
            memcpy(W.host + o_rb + r0, ref_bases + r0, r1 - r0);
            memcpy(W.host + o_ab + q0, alt_bases + q0, q1 - q0);
            good = (r1 == r0 || ok(h, hipMemcpyAsync(W.dev + o_rb + r0, W.host + o_rb + r0, r1 - r0, hipMemcpyHostToDevice, S_in), "H2D sw")) &&
                   (q1 == q0 || ok(h, hipMemcpyAsync(W.dev + o_ab + q0, W.host + o_ab + q0, q1 - q0, hipMemcpyHostToDevice, S_in), "H2D sw")) &&
                   ok(h, hipEventRecord(W.ev_in[c], S_in), "hipEventRecord") && ok(h, hipStreamWaitEvent(S, W.ev_in[c], 0), "hipStreamWaitEvent");
            if (!good || a1 == a0) continue;
            p.a_begin = a0;
            p.n_alignments = a1;
            const size_t workers = std::min<size_t>(max_workers, ((size_t)(a1 - a0) + gpb - 1) / gpb);
            (void)hipEventRecord(W.ev_k0[c], S);
            good = ok(h, launch_sw(L, K, p, (uint32_t)workers, lds, S), "phmm_sw_align_kernel");
            (void)hipEventRecord(W.ev_k1[c], S);
        }
        // (while the device works) what the kernels store per alignment: (rows + L - 1) steps x L lanes x flag words per strip
        W.last_backtrack_bytes = 0;
        for (uint32_t a = 0; a < n_alignments; ++a)
            W.last_backtrack_bytes += (uint64_t)((alt_off[a + 1] - alt_off[a] + strip_cols - 1) / strip_cols) *
                                      (ref_off[a + 1] - ref_off[a] + L - 1ull) * L * flag_words * 4ull;
        // Results come back piece by piece, on a stream of their own: a piece's D2H is issued once the host has seen
        // its kernel finish (a copy that waits in the queue for a kernel holds back the H2D copies behind it, DESIGN.md
        // section 9), and is unpacked into the caller's arrays while the later pieces compute.
        hipStream_t S_out = h->streams[2];
        auto unpack = [&](int c) {
            const uint32_t a0 = cut[c], a1 = cut[c + 1];
            if (!ok(h, hipEventSynchronize(W.ev_out[c]), "sync(sw results)")) return false;
            memcpy(n_cigar + a0, W.host + o_nc + 4ull * a0, 4ull * (a1 - a0));
            memcpy(alignment_offset + a0, W.host + o_of + 4ull * a0, 4ull * (a1 - a0));
            if (cigar_off[a1] > cigar_off[a0]) memcpy(cigar + cigar_off[a0], W.host + o_cg + 4ull * cigar_off[a0], 4ull * (cigar_off[a1] - cigar_off[a0]));
            return true;
        };
        int prev = -1;
        for (int c = 0; c < n_chunks && good; ++c) {
            const uint32_t a0 = cut[c], a1 = cut[c + 1];
            if (a1 == a0) continue;
            const uint64_t g0 = cigar_off[a0], g1 = cigar_off[a1];
            good = ok(h, hipEventSynchronize(W.ev_k1[c]), "sync(sw kernel)") &&
                   ok(h, hipMemcpyAsync(W.host + o_nc + 4ull * a0, W.dev + o_nc + 4ull * a0, 4ull * (a1 - a0), hipMemcpyDeviceToHost, S_out), "D2H sw") &&
                   ok(h, hipMemcpyAsync(W.host + o_of + 4ull * a0, W.dev + o_of + 4ull * a0, 4ull * (a1 - a0), hipMemcpyDeviceToHost, S_out), "D2H sw") &&
                   (g1 == g0 || ok(h, hipMemcpyAsync(W.host + o_cg + 4ull * g0, W.dev + o_cg + 4ull * g0, 4ull * (g1 - g0), hipMemcpyDeviceToHost, S_out), "D2H sw")) &&
                   ok(h, hipEventRecord(W.ev_out[c], S_out), "hipEventRecord");
            if (good && prev >= 0) good = unpack(prev);
            prev = c;
        }
        good = good && ok(h, hipMemcpyAsync(W.host + o_st, W.dev, 256, hipMemcpyDeviceToHost, S_out), "D2H sw");
        if (good && prev >= 0) good = unpack(prev);
        good = good && ok(h, hipStreamSynchronize(S_out), "sync(sw)");
        if (!good) {
            (void)hipStreamSynchronize(S_in);
            (void)hipStreamSynchronize(S);
            (void)hipStreamSynchronize(S_out);
            return PHMM_ERR_HIP;
        }
        W.last_kernel_us = 0;
        for (int c = 0; c < n_chunks; ++c) {
            float ms = 0.f;
            if (cut[c + 1] > cut[c] && hipEventElapsedTime(&ms, W.ev_k0[c], W.ev_k1[c]) == hipSuccess) W.last_kernel_us += (uint64_t)(ms * 1e3f);
        }
        {
            const uint32_t *st = (const uint32_t *)(W.host + o_st + 64);  // [2], [3]: shader clocks / 100 MHz ticks of the last kernel's block 0
            W.last_clock_mhz = st[3] ? (uint64_t)((double)st[2] * 100.0 / (double)st[3]) : 0;
        }
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
