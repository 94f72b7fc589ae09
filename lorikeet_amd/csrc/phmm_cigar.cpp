// phmm_project_to_reference (include/phmm.h): host side -- validation, staging, the workspace, status.  The CIGAR algebra
// itself runs on the device (phmm_cigar_kernels.hip); there is no CPU path here.
#include <algorithm>
#include <cstring>
#include <new>
#include <string>

#include "phmm_cigar_internal.hpp"
#include "phmm_host.hpp"

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

int fail(phmm_handle *h, const char *msg) {
    h->err = std::string("phmm_project_to_reference: ") + msg;
    return h->err_code = PHMM_ERR_INVALID_ARG;
}

}  // namespace

extern "C" int phmm_project_to_reference(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off,
                                         const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases,
                                         const uint32_t *hap_off, const uint8_t *hap_bases, const int32_t *region_ref_hap,
                                         const uint64_t *region_reference_start, const uint32_t *hap_cigar_off,
                                         const uint32_t *hap_cigar, const uint32_t *hap_start_wrt_ref, const int32_t *best_allele,
                                         const uint64_t *sw_cigar_off, const uint32_t *sw_cigar, const uint32_t *n_sw_cigar,
                                         const int32_t *sw_offset, const uint32_t *orig_cigar_off, const uint32_t *orig_cigar,
                                         const uint64_t *out_cigar_off, uint32_t *out_cigar, uint32_t *n_out_cigar,
                                         int64_t *new_pos, int32_t *status) {
    if (!h) return PHMM_ERR_INVALID_ARG;
    try {
        phmm_host::latch_slot0(h);
        h->err_code = PHMM_OK;
        if (!n_regions) return PHMM_OK;
        if (!region_read_off || !region_hap_off || !region_ref_hap || !region_reference_start) return fail(h, "null array");
        if (region_read_off[0] != 0 || region_hap_off[0] != 0) return fail(h, "offset arrays must start at 0");
        for (uint32_t g = 0; g < n_regions; ++g) {
            if (region_read_off[g + 1] < region_read_off[g] || region_hap_off[g + 1] < region_hap_off[g]) return fail(h, "offsets not monotonic");
            const uint32_t nh = region_hap_off[g + 1] - region_hap_off[g];
            if (region_read_off[g + 1] > region_read_off[g] && (region_ref_hap[g] < 0 || (uint32_t)region_ref_hap[g] >= nh))
                return fail(h, "every region with reads needs its reference haplotype (region_ref_hap inside the region)");
        }
        const uint32_t n_reads = region_read_off[n_regions], n_haps = region_hap_off[n_regions];
        if (!n_reads) return PHMM_OK;
        if (!read_off || !hap_off || !hap_cigar_off || !hap_start_wrt_ref || !best_allele || !sw_cigar_off || !n_sw_cigar || !sw_offset ||
            !orig_cigar_off || !out_cigar_off || !n_out_cigar || !new_pos || !status)
            return fail(h, "null array");
        if (read_off[0] != 0 || hap_off[0] != 0 || hap_cigar_off[0] != 0 || sw_cigar_off[0] != 0 || orig_cigar_off[0] != 0 || out_cigar_off[0] != 0)
            return fail(h, "offset arrays must start at 0");
        uint32_t max_sw = 0, max_hc = 0, max_oc = 0;
        for (uint32_t a = 0; a < n_haps; ++a) {
            if (hap_off[a + 1] < hap_off[a] || hap_cigar_off[a + 1] < hap_cigar_off[a]) return fail(h, "offsets not monotonic");
            max_hc = std::max(max_hc, hap_cigar_off[a + 1] - hap_cigar_off[a]);
        }
        for (uint32_t r = 0; r < n_reads; ++r) {
            if (read_off[r + 1] < read_off[r] || sw_cigar_off[r + 1] < sw_cigar_off[r] || orig_cigar_off[r + 1] < orig_cigar_off[r] ||
                out_cigar_off[r + 1] < out_cigar_off[r])
                return fail(h, "offsets not monotonic");
            if (n_sw_cigar[r] > sw_cigar_off[r + 1] - sw_cigar_off[r]) return fail(h, "n_sw_cigar exceeds the read's slot");
            max_sw = std::max(max_sw, n_sw_cigar[r]);
            max_oc = std::max(max_oc, orig_cigar_off[r + 1] - orig_cigar_off[r]);
        }
        const size_t rb = read_off[n_reads], hb = hap_off[n_haps], n_hc = hap_cigar_off[n_haps], n_oc = orig_cigar_off[n_reads];
        const uint64_t n_sw = sw_cigar_off[n_reads], n_out = out_cigar_off[n_reads];
        if ((rb && !read_bases) || (hb && !hap_bases) || (n_hc && !hap_cigar) || (n_sw && !sw_cigar) || (n_oc && !orig_cigar) || (n_out && !out_cigar))
            return fail(h, "null array");
        // elements a lane may hold at once: the padded haplotype cigar; the projection (at most one element per pair of
        // input elements); four per element of that in left_align_indels plus two
        const uint32_t capacity = 4 * (max_sw + max_hc + 2) + 8;
        (void)max_oc;

        DevGuard dg(h->device);
        phmm_handle::SwWork &W = h->swork;
        hipStream_t S = h->streams[0];
        // ---- staging: inputs, then [flags | status | n_out | new_pos | out cigar] ----------------------------------------
        size_t o = 0;
        auto place = [&](size_t bytes) {
            const size_t at = o;
            o += up256(bytes);
            return at;
        };
        const size_t o_rro = place(4ull * (n_regions + 1)), o_rho = place(4ull * (n_regions + 1)), o_ro = place(4ull * (n_reads + 1)),
                     o_rb = place(rb), o_ho = place(4ull * (n_haps + 1)), o_hb = place(hb), o_rrh = place(4ull * n_regions),
                     o_rs = place(8ull * n_regions), o_hco = place(4ull * (n_haps + 1)), o_hc = place(4ull * n_hc), o_hs = place(4ull * n_haps),
                     o_ba = place(4ull * n_reads), o_swo = place(8ull * (n_reads + 1)), o_sw = place(4ull * n_sw), o_nsw = place(4ull * n_reads),
                     o_so = place(4ull * n_reads), o_oco = place(4ull * (n_reads + 1)), o_oc = place(4ull * n_oc), o_oo = place(8ull * (n_reads + 1)),
                     in_bytes = o;
        const size_t o_fl = place(256), o_st = place(4ull * n_reads), o_no = place(4ull * n_reads), o_np = place(8ull * n_reads),
                     o_out = place(4ull * n_out), total = o;
        if (W.cap < total) {
            for (int i = 0; i < 3; ++i) (void)hipStreamSynchronize(h->streams[i]);
            if (W.dev) (void)hipFree(W.dev);
            if (W.host) (void)hipHostFree(W.host);
            W.dev = W.host = W.host_dev = nullptr;
            W.cap = 0;
            const size_t cap = std::max<size_t>(total + total / 2, 1 << 20);
            if (!ok(h, hipMalloc((void **)&W.dev, cap), "hipMalloc(project staging)") ||
                !ok(h, hipHostMalloc((void **)&W.host, cap, hipHostMallocDefault), "hipHostMalloc(project staging)"))
                return PHMM_ERR_HIP;
            W.cap = cap;
        }
        const size_t ws_bytes = (size_t)n_reads * 4 * capacity * 4;  // the lanes' builders live in the Smith-Waterman slab
        if (W.slab_bytes < ws_bytes) {
            (void)hipStreamSynchronize(S);
            if (W.slab) (void)hipFree(W.slab);
            W.slab = nullptr;
            W.slab_bytes = 0;
            if (!ok(h, hipMalloc((void **)&W.slab, ws_bytes), "hipMalloc(project workspace)")) return PHMM_ERR_HIP;
            W.slab_bytes = ws_bytes;
        }
        auto put = [&](size_t at, const void *src, size_t bytes) {
            if (bytes) memcpy(W.host + at, src, bytes);
        };
        put(o_rro, region_read_off, 4ull * (n_regions + 1));
        put(o_rho, region_hap_off, 4ull * (n_regions + 1));
        put(o_ro, read_off, 4ull * (n_reads + 1));
        put(o_rb, read_bases, rb);
        put(o_ho, hap_off, 4ull * (n_haps + 1));
        put(o_hb, hap_bases, hb);
        put(o_rrh, region_ref_hap, 4ull * n_regions);
        put(o_rs, region_reference_start, 8ull * n_regions);
        put(o_hco, hap_cigar_off, 4ull * (n_haps + 1));
        put(o_hc, hap_cigar, 4ull * n_hc);
        put(o_hs, hap_start_wrt_ref, 4ull * n_haps);
        put(o_ba, best_allele, 4ull * n_reads);
        put(o_swo, sw_cigar_off, 8ull * (n_reads + 1));
        put(o_sw, sw_cigar, 4ull * n_sw);
        put(o_nsw, n_sw_cigar, 4ull * n_reads);
        put(o_so, sw_offset, 4ull * n_reads);
        put(o_oco, orig_cigar_off, 4ull * (n_reads + 1));
        put(o_oc, orig_cigar, 4ull * n_oc);
        put(o_oo, out_cigar_off, 8ull * (n_reads + 1));
        memset(W.host + o_fl, 0, 256);
        ProjectParams p{};
        p.n_reads = n_reads;
        p.n_regions = n_regions;
        p.region_read_off = (const uint32_t *)(W.dev + o_rro);
        p.region_hap_off = (const uint32_t *)(W.dev + o_rho);
        p.read_off = (const uint32_t *)(W.dev + o_ro);
        p.read_bases = (const uint8_t *)(W.dev + o_rb);
        p.hap_off = (const uint32_t *)(W.dev + o_ho);
        p.hap_bases = (const uint8_t *)(W.dev + o_hb);
        p.region_ref_hap = (const int32_t *)(W.dev + o_rrh);
        p.region_reference_start = (const uint64_t *)(W.dev + o_rs);
        p.hap_cigar_off = (const uint32_t *)(W.dev + o_hco);
        p.hap_cigar = (const uint32_t *)(W.dev + o_hc);
        p.hap_start_wrt_ref = (const uint32_t *)(W.dev + o_hs);
        p.best_allele = (const int32_t *)(W.dev + o_ba);
        p.sw_cigar_off = (const uint64_t *)(W.dev + o_swo);
        p.sw_cigar = (const uint32_t *)(W.dev + o_sw);
        p.n_sw_cigar = (const uint32_t *)(W.dev + o_nsw);
        p.sw_offset = (const int32_t *)(W.dev + o_so);
        p.orig_cigar_off = (const uint32_t *)(W.dev + o_oco);
        p.orig_cigar = (const uint32_t *)(W.dev + o_oc);
        p.out_cigar_off = (const uint64_t *)(W.dev + o_oo);
        p.out_cigar = (uint32_t *)(W.dev + o_out);
        p.n_out_cigar = (uint32_t *)(W.dev + o_no);
        p.new_pos = (int64_t *)(W.dev + o_np);
        p.status = (int32_t *)(W.dev + o_st);
        p.flags = (uint32_t *)(W.dev + o_fl);
        p.workspace = W.slab;
        p.capacity = capacity;
        if (!ok(h, hipMemcpyAsync(W.dev, W.host, in_bytes + 256, hipMemcpyHostToDevice, S), "H2D project") ||
            !ok(h, launch_project(p, S), "phmm_project_kernel") ||
            !ok(h, hipMemcpyAsync(W.host + o_fl, W.dev + o_fl, total - o_fl, hipMemcpyDeviceToHost, S), "D2H project") ||
            !ok(h, hipStreamSynchronize(S), "sync(project)"))
            return PHMM_ERR_HIP;
        memcpy(status, W.host + o_st, 4ull * n_reads);
        memcpy(n_out_cigar, W.host + o_no, 4ull * n_reads);
        memcpy(new_pos, W.host + o_np, 8ull * n_reads);
        if (n_out) memcpy(out_cigar, W.host + o_out, 4ull * n_out);
        if (*(const uint32_t *)(W.host + o_fl) & 1u) {
            h->err = "phmm_project_to_reference: a CIGAR needs more elements than its slot holds (n_out_cigar has the sizes)";
            return h->err_code = PHMM_ERR_CIGAR_CAPACITY;
        }
        return PHMM_OK;
    } catch (const std::bad_alloc &) {
        h->err = "phmm_project_to_reference: out of host memory";
        return h->err_code = PHMM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        h->err = std::string("phmm_project_to_reference: ") + e.what();
        return h->err_code = PHMM_ERR_INTERNAL;
    }
}
