// phmm_sw_align / phmm_sw_align_indexed / phmm_best_alleles / phmm_realign_to_best (include/phmm.h): host side of the
// Smith-Waterman aligner and of the best-allele step in front of it -- validation, staging, worker geometry, the
// pipeline of pieces, status.  The work itself (matrix, backtrack, CIGAR: phmm_sw_kernels.hip; best allele:
// phmm_engine_kernels.hip) runs on the device; there is no CPU path here.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "phmm_cigar_internal.hpp"
#include "phmm_host.hpp"
#include "phmm_sw_internal.hpp"

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

// The best-allele step (phmm_best_alleles, and in front of the alignments of phmm_realign_to_best).
struct BestJob {
    uint32_t n_regions = 0, n_reads = 0, n_haps = 0;
    const uint32_t *region_read_off = nullptr, *region_hap_off = nullptr;
    const uint64_t *out_off = nullptr;
    const double *likelihoods = nullptr;
    const uint8_t *keep = nullptr;
    const int32_t *priority = nullptr;
    double threshold = 0.2;
    int32_t *best_allele = nullptr;
    double *likelihood = nullptr, *confidence = nullptr;
};

// The projection onto the reference behind the alignments (phmm_realign_reads): the alignments never leave the device.
struct ProjJob {
    const int32_t *region_ref_hap = nullptr;
    const uint64_t *region_reference_start = nullptr;
    const uint32_t *hap_cigar_off = nullptr, *hap_cigar = nullptr, *hap_start_wrt_ref = nullptr;
    const uint32_t *orig_cigar_off = nullptr, *orig_cigar = nullptr;
    const uint64_t *out_cigar_off = nullptr;
    uint32_t *out_cigar = nullptr, *n_out_cigar = nullptr;
    int64_t *new_pos = nullptr;
    int32_t *status = nullptr;
    uint32_t sw_capacity = 24;  // CIGAR elements reserved per alignment on the device (grown and redone when one needs more)
    uint32_t max_hap_cigar = 0;
};

// For a caller inside the library that works on the alignments where they lie (phmm_calculate_cigar): they stay on the device.
struct SwDeviceView {
    uint32_t sw_capacity = 32;   // in: CIGAR elements reserved per alignment
    size_t extra_bytes = 0;      // in: room behind everything else in the staging buffers, for the caller's own results
    const uint32_t *ref_off = nullptr, *alt_off = nullptr;  // out: device pointers
    const uint8_t *ref_bases = nullptr, *alt_bases = nullptr;
    const uint64_t *cigar_off = nullptr;
    const uint32_t *cigar = nullptr, *n_cigar = nullptr;
    const int32_t *alignment_offset = nullptr;
    char *extra_dev = nullptr, *extra_host = nullptr;
};

// One batch of alignments: alignment a pairs alternate sequence a with reference ref_index[a] (or a when there is no index;
// or the best allele's haplotype when `best` runs in front).
struct SwJob {
    const char *who = "phmm_sw_align";
    uint32_t n_alignments = 0, n_refs = 0;
    const uint32_t *ref_off = nullptr, *alt_off = nullptr, *ref_index = nullptr;
    const uint8_t *ref_bases = nullptr, *alt_bases = nullptr;
    const phmm_sw_parameters *params = nullptr;
    int strategy = 0;
    const uint64_t *cigar_off = nullptr;
    uint32_t *cigar = nullptr, *n_cigar = nullptr;
    int32_t *alignment_offset = nullptr;
    const BestJob *best = nullptr;
    const ProjJob *proj = nullptr;  // with it: cigar / n_cigar / alignment_offset stay on the device (the pointers above are unused)
    SwDeviceView *view = nullptr;   // likewise, for a caller that launches its own kernel on them afterwards
    uint32_t *sw_capacity_needed = nullptr;  // out: the largest alignment CIGAR when the reserved slots were too small
};

int fail(phmm_handle *h, const std::string &msg) {
    h->err = msg;
    return h->err_code = PHMM_ERR_INVALID_ARG;
}

int check_best(phmm_handle *h, const char *who, const BestJob &b) {
    const std::string w(who);
    if (!b.n_regions) return PHMM_OK;
    if (!b.region_read_off || !b.region_hap_off || !b.out_off) return fail(h, w + ": null array");
    if (b.region_read_off[0] != 0 || b.region_hap_off[0] != 0) return fail(h, w + ": offset arrays must start at 0");
    for (uint32_t g = 0; g < b.n_regions; ++g) {
        if (b.region_read_off[g + 1] < b.region_read_off[g] || b.region_hap_off[g + 1] < b.region_hap_off[g] || b.out_off[g + 1] < b.out_off[g])
            return fail(h, w + ": offsets not monotonic");
        const uint64_t need = (uint64_t)(b.region_read_off[g + 1] - b.region_read_off[g]) * (b.region_hap_off[g + 1] - b.region_hap_off[g]);
        if (b.out_off[g + 1] - b.out_off[g] < need) return fail(h, w + ": out_off leaves too little room for a region's matrix");
    }
    if (b.n_reads && (!b.best_allele || !b.likelihood || !b.confidence || (b.out_off[b.n_regions] && !b.likelihoods)))
        return fail(h, w + ": null array");
    if (!(b.threshold >= 0.0)) return fail(h, w + ": the informative threshold must be a non-negative number");
    return PHMM_OK;
}

// staging layout of the best-allele step inside a buffer, starting at `base`: inputs up to `best`, results up to `end`
struct BestLayout {
    size_t rro, rho, oo, lk, keep, pri, best, olk, conf, end;
    BestLayout(const BestJob *b, size_t base) {
        const size_t ng = b ? b->n_regions : 0, nr = b ? b->n_reads : 0, nh = b ? b->n_haps : 0, no = b && ng ? b->out_off[ng] : 0;
        rro = base;
        rho = rro + (b ? up256(4 * (ng + 1)) : 0);
        oo = rho + (b ? up256(4 * (ng + 1)) : 0);
        lk = oo + (b ? up256(8 * (ng + 1)) : 0);
        keep = lk + up256(8 * no);
        pri = keep + (b && b->keep ? up256(nr) : 0);
        best = pri + (b && b->priority ? up256(4 * nh) : 0);
        olk = best + up256(4 * nr);
        conf = olk + up256(8 * nr);
        end = conf + up256(8 * nr);
    }
};

// (`with_likelihoods` = false: a call in pieces stages the matrix piece by piece, see sw_run)
void stage_best(const BestJob &b, const BestLayout &L, char *host, bool with_likelihoods = true) {
    memcpy(host + L.rro, b.region_read_off, 4ull * (b.n_regions + 1));
    memcpy(host + L.rho, b.region_hap_off, 4ull * (b.n_regions + 1));
    memcpy(host + L.oo, b.out_off, 8ull * (b.n_regions + 1));
    if (with_likelihoods && b.out_off[b.n_regions]) memcpy(host + L.lk, b.likelihoods, 8ull * b.out_off[b.n_regions]);
    if (b.keep) memcpy(host + L.keep, b.keep, b.n_reads);
    if (b.priority) memcpy(host + L.pri, b.priority, 4ull * b.n_haps);
}

// results of at most this many bytes are stored into the pinned mirror by the kernels themselves (sw_run)
constexpr size_t kZeroCopyResultBytes = 64u << 10;
// ... and inputs of at most this many are fetched from it by a kernel instead of the copy engine
constexpr size_t kStageInBytes = 256u << 10;

BestParams best_params(const BestJob &b, const BestLayout &L, char *dev, char *out, uint32_t *d_ref_index) {
    BestParams p{};
    p.n_reads = b.n_reads;
    p.n_regions = b.n_regions;
    p.region_read_off = (const uint32_t *)(dev + L.rro);
    p.region_hap_off = (const uint32_t *)(dev + L.rho);
    p.out_off = (const uint64_t *)(dev + L.oo);
    p.likelihoods = (const double *)(dev + L.lk);
    p.keep = b.keep ? (const uint8_t *)(dev + L.keep) : nullptr;
    p.priority = b.priority ? (const int32_t *)(dev + L.pri) : nullptr;
    p.threshold = b.threshold;
    p.best_allele = (int32_t *)(out + L.best);
    p.likelihood = (double *)(out + L.olk);
    p.confidence = (double *)(out + L.conf);
    p.ref_index = d_ref_index;
    return p;
}

bool grow_staging(phmm_handle *h, size_t total) {
    phmm_handle::SwWork &W = h->swork;
    if (W.cap >= total) return true;
    for (int i = 0; i < 3; ++i) (void)hipStreamSynchronize(h->streams[i]);
    if (W.dev) (void)hipFree(W.dev);
    if (W.host) (void)hipHostFree(W.host);
    W.dev = W.host = W.host_dev = nullptr;
    W.cap = 0;
    const size_t cap = std::max<size_t>(total + total / 2, 1 << 20);
    if (!ok(h, hipMalloc((void **)&W.dev, cap), "hipMalloc(sw staging)") ||
        !ok(h, hipHostMalloc((void **)&W.host, cap, hipHostMallocDefault), "hipHostMalloc(sw staging)"))
        return false;
    W.cap = cap;
    return true;
}

}  // namespace

namespace phmm_host {

// Worker geometry of one batch of alignments (shared by sw_run and the per-region pipeline, phmm_region.cpp).
int sw_plan(phmm_handle *h, const std::string &who, uint32_t n_alignments, uint32_t max_ref, uint32_t max_alt,
            const phmm_sw_parameters *params, SwGeometry *G) {
    // The scaled kernels carry scores times four with a two-bit tag and leave out the reference's clamp at -1e8
    // (MATRIX_MIN_CUTOFF): exact as long as no score can get near that clamp.  Beyond that the wide instance takes over
    // (scores as they are, the reference's comparisons and its clamp) -- up to where the reference's own 32-bit sums
    // (low_init_value = i32::MIN / 2 plus one gap extension per row, :137, :203) would overflow, which it does not survive
    // either (a panic in debug builds, wrapped values in release builds).
    bool wide = false;
    {
        const int64_t big = std::max(std::max(std::llabs((long long)params->match_value), std::llabs((long long)params->mismatch_penalty)),
                                     std::max(std::llabs((long long)params->gap_open_penalty), std::llabs((long long)params->gap_extend_penalty)));
        const int64_t reach = big * ((int64_t)max_ref + max_alt + 2);
        if (reach >= 1000000000)
            return fail(h, who + ": parameters too large for these sequence lengths (|weight| x (ref + alt) must stay below 1e9: "
                                 "beyond that the reference's own 32-bit arithmetic overflows)");
        wide = reach >= 100000000;
    }
    // ---- geometry ---------------------------------------------------------------------------------------------
    // L lanes per alignment, K columns per lane, so that one strip of L x K columns covers the longest alternate sequence
    // (more than 512 columns take several strips of 512).  Throughput wants few lanes per alignment -- eight alignments
    // per wave lose the fewest steps to the skew and spread the per-step work over the most cells -- latency wants many:
    // a call that cannot fill the chip anyway (a round of the persistent blocks is 32 768 alignments at eight lanes) gets
    // 16, 32 or 64 lanes per alignment, i.e. more waves with less work each (one region of 128 reads: 305 us at 8 lanes,
    // 220 at 16).
    const int force_L = wide ? 16 : h->sw.sw_lanes;
    // (64 lanes up to 2 048 alignments where the sweep along the alternate applies, below: 16 regions of 128 reads 248 -> 228 us)
    const uint32_t most_at_64 = h->sw.sw_transpose != 0 && max_ref > max_alt && max_ref <= 512 ? 2048 : 1024;
    int L = force_L ? force_L : max_alt <= 8 * 20 && n_alignments >= 32768 ? 8 : max_alt > 512 || n_alignments > 4096 ? 16 : n_alignments > most_at_64 ? 32 : 64;
    const int *ks = L == 8 ? kSwK8 : L == 16 ? kSwK16 : L == 32 ? kSwK32 : kSwK64;
    const int nks = L == 8 ? kNumSwK8 : L == 16 ? kNumSwK16 : L == 32 ? kNumSwK32 : kNumSwK64;
    int K = ks[nks - 1];
    for (int i = nks - 1; i >= 0; --i)
        if ((size_t)ks[i] * L >= max_alt) K = ks[i];
    if (wide) K = 16;  // (the one wide instance)
    // A small call whose references are longer than its alternates (reads against their haplotypes) sweeps along the
    // ALTERNATE instead, the reference's rows shared out over the 64 lanes: fewer steps of more cells each, and a step's
    // fixed cost (~50 instructions next to 16 per cell) is what a lone wave per SIMD feels -- 150 x 300: 350 steps of
    // three cells against 210 of five.  One strip only.
    bool transposed = false;
    if (!wide && L == 64 && h->sw.sw_transpose != 0 && max_ref <= 64u * (uint32_t)kSwK64T[kNumSwK64T - 1]) {
        int KT = kSwK64T[kNumSwK64T - 1];
        for (int i = kNumSwK64T - 1; i >= 0; --i)
            if ((size_t)kSwK64T[i] * 64 >= max_ref) KT = kSwK64T[i];
        auto cost = [](size_t sweep, size_t across, int k) { return (sweep + (across + k - 1) / k) * (16ull * k + 50); };
        const bool one_strip = (size_t)K * 64 >= max_alt;
        if (h->sw.sw_transpose > 0 || !one_strip || cost(max_alt, max_ref, KT) < cost(max_ref, max_alt, K)) {
            transposed = true;
            K = KT;
        }
    }
    size_t strip_cols, strips, lds_group, gpb, lds, ext_stride = 0;
    const size_t lds_ref = (max_ref + 15) / 16 * 16, lds_alt = (max_alt + 15) / 16 * 16;
    auto layout = [&]() {
        strip_cols = (size_t)L * K;
        strips = transposed ? 1 : (max_alt + strip_cols - 1) / strip_cols;
        // per alignment: the two sequences, the bottom row, and (several strips only) the strip edge, two i32 per row
        lds_group = (lds_ref + lds_alt + 4ull * (max_alt + 1) + (strips > 1 ? 8ull * (max_ref + 1) : 0) + 15) / 16 * 16;
        // 64 / L alignments share a wave; sequences so long that they do not fit a block's LDS together get the wave to themselves
        gpb = (64 / L) * lds_group <= 160 * 1024 ? 64 / L : 1;
        lds = gpb * lds_group;
    };
    layout();
    if (lds > 160 * 1024) {
        // Beyond ~8 000 bases the bottom row and the strip edges (4 and 8 bytes per base) no longer fit next to the sequences:
        // they move to device memory, one slice per block -- slower per step, but the reference aligns any lengths
        // (smith_waterman_aligner.rs:47-107) and so does this.  The sequences themselves stay in LDS (up to ~80 000 bases each).
        // One instance does it: 16 lanes x 32 columns (16 x 16 for wide weights), one alignment per block.
        L = 16;
        K = wide ? 16 : 32;
        transposed = false;
        layout();
        gpb = 1;
        lds = lds_ref + lds_alt;
        ext_stride = (4ull * (max_alt + 1) + 8ull * (max_ref + 1) + 255) / 256 * 256;
        if (lds > 160 * 1024) return fail(h, who + ": sequences too long for the LDS staging (about 80 000 bases each)");
    }
    const int variant = (wide ? phmm::SW_WIDE : 0) | (ext_stride ? phmm::SW_EXT : 0);
    // persistent blocks (one wave each, `gpb` alignments at a time): exactly what the chip holds at once -- more would
    // queue behind the first ones and leave the last round ragged -- capped by the work and by 6 GB of backtrack storage
    int per_cu;
    {
        const uint64_t key = (uint64_t)L << 56 | (uint64_t)K << 48 | (uint64_t)transposed << 47 | (uint64_t)variant << 45 | (uint64_t)lds;
        auto it = h->swork.blocks_per_cu.find(key);
        if (it == h->swork.blocks_per_cu.end()) {
            if (h->swork.blocks_per_cu.size() >= 4096) h->swork.blocks_per_cu.clear();  // (LDS sizes follow the longest sequences of a call)
            it = h->swork.blocks_per_cu.emplace(key, sw_blocks_per_cu(L, K, lds, transposed, variant)).first;
        }
        per_cu = it->second;
    }
    if (per_cu <= 0) {
        h->err = who + ": the kernel does not fit a compute unit";
        return h->err_code = PHMM_ERR_INTERNAL;
    }
    // backtrack flags per block: strips x (rows + L - 1) steps x sw_flag_words(K) ~ K / 8 dwords x 64 lanes (four bits per cell)
    const size_t flag_words = (size_t)sw_flag_words(K);
    const size_t slab_stride = strips * (size_t)((transposed ? std::max(max_ref, max_alt) : max_ref) + L) * flag_words * 64;
    const size_t max_workers = std::max<size_t>(1, std::min<size_t>(256 * (size_t)per_cu, (6ull << 30) / (slab_stride * 4)));
    G->L = L;
    G->K = K;
    G->transposed = transposed;
    G->wide = wide;
    G->variant = variant;
    G->strips = strips;
    G->lds_ref = lds_ref;
    G->lds_alt = lds_alt;
    G->lds_group = lds_group;
    G->gpb = gpb;
    G->lds = lds;
    G->per_cu = per_cu;
    G->flag_words = flag_words;
    G->slab_stride = slab_stride;
    G->max_workers = max_workers;
    G->ext_stride = ext_stride;
    return PHMM_OK;
}

}  // namespace phmm_host

namespace {

int sw_run(phmm_handle *h, const SwJob &J) {
    const std::string who(J.who);
    h->err_code = PHMM_OK;
    const phmm_sw_parameters *params = J.params;
    const uint32_t n_alignments = J.n_alignments, n_refs = J.n_refs;
    const uint32_t *ref_off = J.ref_off, *alt_off = J.alt_off;
    if (!params) return fail(h, who + ": null parameters");
    if (J.strategy < PHMM_SW_SOFTCLIP || J.strategy > PHMM_SW_IGNORE) return fail(h, who + ": unknown overhang strategy");
    if (J.best) {
        const int st = check_best(h, J.who, *J.best);
        if (st != PHMM_OK) return st;
    }
    if (!n_alignments) return PHMM_OK;
    const ProjJob *PJ = J.proj;
    const bool on_device = PJ || J.view;  // the alignments are consumed where they lie
    if (!ref_off || !alt_off || (!on_device && (!J.cigar_off || !J.n_cigar || !J.alignment_offset))) return fail(h, who + ": null array");
    std::vector<uint64_t> own_cigar_off;  // ... and keep to slots of the library's own
    if (on_device) {
        const uint64_t slot = PJ ? PJ->sw_capacity : J.view->sw_capacity;
        own_cigar_off.resize((size_t)n_alignments + 1);
        for (size_t a = 0; a <= n_alignments; ++a) own_cigar_off[a] = a * slot;
    }
    const uint64_t *cigar_off = on_device ? own_cigar_off.data() : J.cigar_off;
    if (ref_off[0] != 0 || alt_off[0] != 0 || cigar_off[0] != 0) return fail(h, who + ": offset arrays must start at 0");
    if (PJ) {
        const BestJob &B = *J.best;
        if (!PJ->region_ref_hap || !PJ->region_reference_start || !PJ->hap_cigar_off || !PJ->hap_start_wrt_ref || !PJ->orig_cigar_off ||
            !PJ->out_cigar_off || !PJ->n_out_cigar || !PJ->new_pos || !PJ->status)
            return fail(h, who + ": null array");
        if (PJ->hap_cigar_off[0] != 0 || PJ->orig_cigar_off[0] != 0 || PJ->out_cigar_off[0] != 0) return fail(h, who + ": offset arrays must start at 0");
        for (uint32_t g = 0; g < B.n_regions; ++g)  // (a region without haplotypes has no best alleles: its reads stay as they are)
            if (B.region_read_off[g + 1] > B.region_read_off[g] && B.region_hap_off[g + 1] > B.region_hap_off[g] &&
                (PJ->region_ref_hap[g] < 0 || (uint32_t)PJ->region_ref_hap[g] >= B.region_hap_off[g + 1] - B.region_hap_off[g]))
                return fail(h, who + ": every region with reads and haplotypes needs its reference haplotype (region_ref_hap inside the region)");
        for (uint32_t a = 0; a < B.n_haps; ++a)
            if (PJ->hap_cigar_off[a + 1] < PJ->hap_cigar_off[a]) return fail(h, who + ": offsets not monotonic");
        for (uint32_t r = 0; r < B.n_reads; ++r)
            if (PJ->orig_cigar_off[r + 1] < PJ->orig_cigar_off[r] || PJ->out_cigar_off[r + 1] < PJ->out_cigar_off[r])
                return fail(h, who + ": offsets not monotonic");
        if ((PJ->hap_cigar_off[B.n_haps] && !PJ->hap_cigar) || (PJ->orig_cigar_off[B.n_reads] && !PJ->orig_cigar) ||
            (PJ->out_cigar_off[B.n_reads] && !PJ->out_cigar))
            return fail(h, who + ": null array");
    }
    if (!n_refs) return fail(h, who + ": no reference sequences");
    uint32_t max_ref = 0, max_alt = 0;
    uint64_t max_slot = 0;  // the largest CIGAR slot of the call
    // The reference asserts / panics on empty input (smith_waterman_aligner.rs:65-68, :132-134) -- for the sequences it
    // actually aligns: a reference no alignment names, or the read of a skipped alignment, may be empty.  Where the device
    // chooses the reference (the best allele) the kernel raises the condition for the alignments it meets (SW_STATUS_EMPTY).
    const char *const empty_msg = ": non-empty sequences are required for the Smith-Waterman calculation";
    for (uint32_t r = 0; r < n_refs; ++r) {
        if (ref_off[r + 1] < ref_off[r]) return fail(h, who + ": offsets not monotonic");
        if (ref_off[r + 1] == ref_off[r] && !J.ref_index && !J.best) return fail(h, who + empty_msg);
        max_ref = std::max(max_ref, ref_off[r + 1] - ref_off[r]);
    }
    if (!max_ref) return fail(h, who + empty_msg);
    for (uint32_t a = 0; a < n_alignments; ++a) {
        if (alt_off[a + 1] < alt_off[a] || cigar_off[a + 1] < cigar_off[a]) return fail(h, who + ": offsets not monotonic");
        if (J.ref_index && J.ref_index[a] != SW_NO_REFERENCE && J.ref_index[a] >= n_refs) return fail(h, who + ": reference index out of range");
        const bool skipped = J.ref_index && J.ref_index[a] == SW_NO_REFERENCE;
        if (!skipped && !J.best && (alt_off[a + 1] == alt_off[a] || (J.ref_index && ref_off[J.ref_index[a] + 1] == ref_off[J.ref_index[a]])))
            return fail(h, who + empty_msg);
        max_alt = std::max(max_alt, alt_off[a + 1] - alt_off[a]);
        max_slot = std::max<uint64_t>(max_slot, cigar_off[a + 1] - cigar_off[a]);
    }
    if (!max_alt) max_alt = 1;  // (only skipped alignments: nothing will be swept)
    const size_t rb = ref_off[n_refs], ab = alt_off[n_alignments];
    const uint64_t n_cig = cigar_off[n_alignments];
    if (!J.ref_bases || !J.alt_bases || (n_cig && !J.cigar && !on_device)) return fail(h, who + ": null array");
    const bool indexed = J.ref_index || J.best;  // references are shared: they all travel with the first piece

    DevGuard dg(h->device);
    // ---- geometry (sw_plan above) ------------------------------------------------------------------------------
    phmm_host::SwGeometry G;
    {
        const int gst = phmm_host::sw_plan(h, who, n_alignments, max_ref, max_alt, params, &G);
        if (gst != PHMM_OK) return gst;
    }
    const int L = G.L, K = G.K;
    const bool transposed = G.transposed;
    const size_t strip_cols = (size_t)L * K, lds_ref = G.lds_ref, lds_alt = G.lds_alt, lds_group = G.lds_group, gpb = G.gpb, lds = G.lds,
                 flag_words = G.flag_words, slab_stride = G.slab_stride;
    // Two passes (SW_LITE) where gaps are the exception: reads against haplotypes (SoftClip / Ignore), in calls of one piece.
    // A call that found a gap in more than three alignments of ten makes the next fifteen calls of this handle go straight to
    // the full instance.
    bool lite = G.variant == SW_PLAIN && h->sw.sw_lite != 0 && (J.strategy == PHMM_SW_SOFTCLIP || J.strategy == PHMM_SW_IGNORE);
    if (lite && h->sw.sw_lite < 0 && h->swork.lite_skip > 0) {
        h->swork.lite_skip -= 1;
        lite = false;
    }
    const size_t max_workers = G.max_workers;
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
    if (h->sw.trace)
        fprintf(stderr, "%s: %u alignments, <%d,%d>%s%s, %d blocks per CU, %zu workers, %zu rounds in %d pieces\n", J.who, n_alignments, L, K,
                transposed ? " transposed" : "", G.wide ? " wide" : "", G.per_cu, max_workers, rounds, n_chunks);
    size_t most = 0;
    for (int c = 0; c < n_chunks; ++c) most = std::max<size_t>(most, cut[c + 1] - cut[c]);
    const size_t slab_bytes = std::min<size_t>(max_workers, (most + gpb - 1) / gpb) * slab_stride * 4;
    phmm_handle::SwWork &W = h->swork;
    if (G.ext_stride) {  // (giant sequences only)
        const size_t need = std::min<size_t>(max_workers, (most + gpb - 1) / gpb) * G.ext_stride;
        if (W.ext_bytes < need) {
            for (int i = 0; i < 3; ++i) (void)hipStreamSynchronize(h->streams[i]);
            if (W.ext) (void)hipFree(W.ext);
            W.ext = nullptr;
            W.ext_bytes = 0;
            if (!ok(h, hipMalloc((void **)&W.ext, need), "hipMalloc(sw rows)")) return PHMM_ERR_HIP;
            W.ext_bytes = need;
        }
    }
    // one piece: everything in order on one stream, one copy each way, one wait (a region per call is the reference's pattern)
    const bool one_piece = n_chunks == 1;
    // (A second pass per piece costs a whole sweep's latency each, however few alignments it holds: a call in pieces takes two
    // passes on its own accord only where one second pass can serve all pieces -- below.)
    if (h->sw.sw_lite < 0 && !one_piece && (PJ ? false : on_device || max_slot > 64)) lite = false;
    // A call in pieces whose results go to the caller takes ONE second pass behind its last piece: every piece's results are
    // fetched as soon as its first pass is done, and the few alignments the second pass redoes come back gathered (below).
    constexpr uint32_t kPatchMax = 4096;
    const bool deferred = lite && !one_piece && !on_device && max_slot <= 64;  // (the gathered entries are as wide as the widest slot)
    // ... and a call in pieces whose alignments are projected where they lie (phmm_realign_reads) runs its second pass and then
    // the projection of every piece behind the last first pass: nothing of a piece is final before, its results follow then.
    const bool deferred_project = lite && !one_piece && PJ != nullptr;
    hipStream_t S = h->streams[0], S_in = one_piece ? S : h->streams[1];
    if (W.slab_bytes < slab_bytes) {
        (void)hipStreamSynchronize(S);
        if (W.slab) (void)hipFree(W.slab);
        W.slab = nullptr;
        W.slab_bytes = 0;
        if (!ok(h, hipMalloc((void **)&W.slab, slab_bytes), "hipMalloc(sw backtrack)")) return PHMM_ERR_HIP;
        W.slab_bytes = slab_bytes;
    }
    // ---- staging: [status | ref_off | alt_off | cigar_off | ref_index | best-allele inputs, results | ref | alt] in,
    //               [status | n_cigar | offsets | cigar] out
    const size_t o_ro = 256, o_ao = o_ro + up256(4ull * (n_refs + 1)), o_co = o_ao + up256(4ull * (n_alignments + 1)),
                 o_ri = o_co + up256(8ull * (n_alignments + 1)), o_bi = o_ri + (indexed ? up256(4ull * n_alignments) : 0);
    const BestLayout BL(J.best, o_bi);
    // the projection's own inputs (phmm_realign_reads) sit behind the best-allele block, its results behind the alignments'
    const uint32_t pj_regions = PJ ? J.best->n_regions : 0, pj_haps = PJ ? J.best->n_haps : 0;
    const size_t pj_hc = PJ ? PJ->hap_cigar_off[pj_haps] : 0, pj_oc = PJ ? PJ->orig_cigar_off[n_alignments] : 0;
    const uint64_t pj_out = PJ ? PJ->out_cigar_off[n_alignments] : 0;
    const size_t o_pi = J.best ? BL.end : o_bi, p_rrh = o_pi, p_rs = p_rrh + (PJ ? up256(4ull * pj_regions) : 0),
                 p_hco = p_rs + (PJ ? up256(8ull * pj_regions) : 0), p_hc = p_hco + (PJ ? up256(4ull * (pj_haps + 1)) : 0),
                 p_hs = p_hc + (PJ ? up256(4ull * pj_hc) : 0), p_oco = p_hs + (PJ ? up256(4ull * pj_haps) : 0),
                 p_oc = p_oco + (PJ ? up256(4ull * (n_alignments + 1)) : 0), p_oo = p_oc + (PJ ? up256(4ull * pj_oc) : 0),
                 p_end = p_oo + (PJ ? up256(8ull * (n_alignments + 1)) : 0);
    const size_t o_rb = p_end, o_ab = o_rb + up256(rb), in_bytes = o_ab + up256(ab);
    const size_t o_st = in_bytes, o_nc = o_st + 256, o_of = o_nc + up256(4ull * n_alignments),
                 o_cg = o_of + up256(4ull * n_alignments), o_pfl = o_cg + up256(4ull * n_cig);
    const size_t o_pst = o_pfl + (PJ ? 256 : 0), o_pno = o_pst + (PJ ? up256(4ull * n_alignments) : 0),
                 o_ppos = o_pno + (PJ ? up256(4ull * n_alignments) : 0), o_pout = o_ppos + (PJ ? up256(8ull * n_alignments) : 0),
                 o_extra = o_pout + (PJ ? up256(4ull * pj_out) : 0), o_todo = o_extra + (J.view ? up256(J.view->extra_bytes) : 0),
                 o_patch = o_todo + up256(4ull * n_alignments),  // (the list of the tags-only pass: device memory only)
                 patch_bytes = 4ull * kPatchMax * (3 + max_slot), total = o_patch + (deferred ? up256(patch_bytes) : 0);
    if (!grow_staging(h, total)) return PHMM_ERR_HIP;
    // A small call in one piece (a region per call, the reference's pattern): the kernels store the results -- and the
    // status block -- straight into the pinned mirror, so that nothing is copied back (each copy costs the call some
    // 10 us, and the copy engine is what concurrent callers end up queueing for).  What a later kernel reads again (the
    // reference index, the alignments the projection consumes) stays in device memory.
    const size_t result_bytes = (PJ ? o_extra - o_pst : J.view ? 0 : o_pfl - o_nc) + (J.best ? BL.end - BL.best : 0);
    bool zero_copy = one_piece && !h->sw.sw_no_zero_copy && result_bytes <= kZeroCopyResultBytes;
    if (zero_copy && !W.host_dev) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, W.host, 0) == hipSuccess && dp)
            W.host_dev = (char *)dp;
        else
            zero_copy = false;
    }
    char *const out_base = zero_copy ? W.host_dev : W.dev;          // results
    char *const st_base = zero_copy ? W.host_dev + o_st : W.dev;    // the status block
    uint32_t pj_capacity = 0;
    if (PJ) {  // the lanes' builders: see phmm_cigar.cpp
        pj_capacity = 4 * (PJ->sw_capacity + PJ->max_hap_cigar + 2) + 8;
        const size_t ws_bytes = most * 4ull * pj_capacity * 4ull;
        if (W.ws_bytes < ws_bytes) {
            (void)hipStreamSynchronize(S);
            if (W.ws) (void)hipFree(W.ws);
            W.ws = nullptr;
            W.ws_bytes = 0;
            if (!ok(h, hipMalloc((void **)&W.ws, ws_bytes), "hipMalloc(project workspace)")) return PHMM_ERR_HIP;
            W.ws_bytes = ws_bytes;
        }
    }
    constexpr uint32_t kShortList = 1024;
    phmm_host::SwGeometry GS;
    bool have_short = false;
    if (lite && L < 64 && n_alignments > 4 * kShortList) {  // (see the launches below)
        const std::string keep_err = h->err;
        const int keep_code = h->err_code;
        have_short = phmm_host::sw_plan(h, who, kShortList, max_ref, max_alt, params, &GS) == PHMM_OK && GS.variant == SW_PLAIN && GS.L == 64 &&
                     GS.slab_stride * 4 <= W.slab_bytes;
        h->err = keep_err;
        h->err_code = keep_code;
    }
    for (int c = 0; c < n_chunks; ++c)  // (each event under its own check: a failure half way must not leave the piece with null events for good)
        if ((!W.ev_in[c] && !ok(h, hipEventCreateWithFlags(&W.ev_in[c], hipEventDisableTiming), "hipEventCreate")) ||
            (!W.ev_out[c] && !ok(h, hipEventCreateWithFlags(&W.ev_out[c], hipEventDisableTiming), "hipEventCreate")) ||
            (!W.ev_k0[c] && !ok(h, hipEventCreate(&W.ev_k0[c]), "hipEventCreate")) ||
            (!W.ev_k1[c] && !ok(h, hipEventCreate(&W.ev_k1[c]), "hipEventCreate")))
            return PHMM_ERR_HIP;
    SwParams p{};
    p.ref_off = (const uint32_t *)(W.dev + o_ro);
    p.alt_off = (const uint32_t *)(W.dev + o_ao);
    p.cigar_off = (const uint64_t *)(W.dev + o_co);
    p.ref_index = indexed ? (const uint32_t *)(W.dev + o_ri) : nullptr;
    p.ref_bases = (const uint8_t *)(W.dev + o_rb);
    p.alt_bases = (const uint8_t *)(W.dev + o_ab);
    p.w_match = params->match_value;
    p.w_mismatch = params->mismatch_penalty;
    p.w_open = params->gap_open_penalty;
    p.w_extend = params->gap_extend_penalty;
    p.strategy = J.strategy;
    char *const sw_out = on_device ? W.dev : out_base;  // (alignments that a kernel consumes stay on the device)
    p.cigar = (uint32_t *)(sw_out + o_cg);
    p.n_cigar = (uint32_t *)(sw_out + o_nc);
    p.alignment_offset = (int32_t *)(sw_out + o_of);
    p.slab = W.slab;
    p.slab_stride = slab_stride;
    p.status = (uint32_t *)(st_base + 64);
    p.max_ref = max_ref;
    p.max_alt = max_alt;
    p.lds_ref_bytes = (uint32_t)lds_ref;
    p.lds_alt_bytes = (uint32_t)lds_alt;
    p.lds_group_bytes = (uint32_t)lds_group;
    p.groups_per_block = (uint32_t)gpb;
    p.report_clock = (h->sw.trace || h->sw.sw_clock) ? 1u : 0u;  // (measurement only: the status block is this call's own until it returns)
    p.ext = G.ext_stride ? W.ext : nullptr;
    p.ext_stride = G.ext_stride;
    // the offset arrays, the status word, the index and the best-allele inputs travel with the first piece -- and, when
    // the references are shared (reads -> their haplotypes), all the references
    memset(W.host, 0, 256);
    if (zero_copy) memset(W.host + o_st, 0, 256);
    memcpy(W.host + o_ro, ref_off, 4ull * (n_refs + 1));
    memcpy(W.host + o_ao, alt_off, 4ull * (n_alignments + 1));
    memcpy(W.host + o_co, cigar_off, 8ull * (n_alignments + 1));
    if (J.ref_index) memcpy(W.host + o_ri, J.ref_index, 4ull * n_alignments);
    size_t head = J.ref_index ? o_bi : o_ri;
    if (J.best) {
        stage_best(*J.best, BL, W.host, one_piece);
        head = BL.best;
    }
    ProjectParams pp{};
    if (PJ) {
        auto put = [&](size_t at, const void *src, size_t bytes) {
            if (bytes) memcpy(W.host + at, src, bytes);
        };
        put(p_rrh, PJ->region_ref_hap, 4ull * pj_regions);
        put(p_rs, PJ->region_reference_start, 8ull * pj_regions);
        put(p_hco, PJ->hap_cigar_off, 4ull * (pj_haps + 1));
        put(p_hc, PJ->hap_cigar, 4ull * pj_hc);
        put(p_hs, PJ->hap_start_wrt_ref, 4ull * pj_haps);
        put(p_oco, PJ->orig_cigar_off, 4ull * (n_alignments + 1));
        put(p_oc, PJ->orig_cigar, 4ull * pj_oc);
        put(p_oo, PJ->out_cigar_off, 8ull * (n_alignments + 1));
        pp.n_regions = pj_regions;
        pp.region_read_off = (const uint32_t *)(W.dev + BL.rro);
        pp.region_hap_off = (const uint32_t *)(W.dev + BL.rho);
        pp.read_off = p.alt_off;
        pp.read_bases = p.alt_bases;
        pp.hap_off = p.ref_off;
        pp.hap_bases = p.ref_bases;
        pp.region_ref_hap = (const int32_t *)(W.dev + p_rrh);
        pp.region_reference_start = (const uint64_t *)(W.dev + p_rs);
        pp.hap_cigar_off = (const uint32_t *)(W.dev + p_hco);
        pp.hap_cigar = (const uint32_t *)(W.dev + p_hc);
        pp.hap_start_wrt_ref = (const uint32_t *)(W.dev + p_hs);
        pp.best_allele = (const int32_t *)(out_base + BL.best);
        pp.sw_cigar_off = p.cigar_off;
        pp.sw_cigar = p.cigar;
        pp.n_sw_cigar = p.n_cigar;
        pp.sw_offset = p.alignment_offset;
        pp.orig_cigar_off = (const uint32_t *)(W.dev + p_oco);
        pp.orig_cigar = (const uint32_t *)(W.dev + p_oc);
        pp.out_cigar_off = (const uint64_t *)(W.dev + p_oo);
        pp.out_cigar = (uint32_t *)(out_base + o_pout);
        pp.n_out_cigar = (uint32_t *)(out_base + o_pno);
        pp.new_pos = (int64_t *)(out_base + o_ppos);
        pp.status = (int32_t *)(out_base + o_pst);
        pp.flags = (uint32_t *)(st_base + 128);  // in the status block: cleared with it, fetched with it
        pp.workspace = W.ws;
        pp.capacity = pj_capacity;
    }
    if (one_piece) {  // the whole input is one contiguous block of the staging buffer
        memcpy(W.host + o_rb, J.ref_bases, rb);
        memcpy(W.host + o_ab, J.alt_bases, ab);
        head = in_bytes;
    }
    bool good;
    if (J.best && !one_piece) {
        // a call in pieces: everything of the head but the likelihood matrix, which is the bulk of it (8 bytes per read
        // and haplotype) and travels with the pieces -- each piece's rows, then the best alleles of its reads
        good = ok(h, hipMemcpyAsync(W.dev, W.host, BL.lk, hipMemcpyHostToDevice, S_in), "H2D sw") &&
               (BL.best == BL.keep || ok(h, hipMemcpyAsync(W.dev + BL.keep, W.host + BL.keep, BL.best - BL.keep, hipMemcpyHostToDevice, S_in), "H2D sw"));
    } else {
        // (a small call's inputs are fetched by a kernel: no copy engine, no cross-engine dependency for the launches behind it)
        good = zero_copy && head <= kStageInBytes ? ok(h, launch_stage_in(W.host_dev, W.dev, head, S_in), "phmm_stage_in_kernel")
                                                  : ok(h, hipMemcpyAsync(W.dev, W.host, head, hipMemcpyHostToDevice, S_in), "H2D sw");
    }
    if (good && J.best && one_piece)  // the reads' best alleles become the index of their references, on the device
        good = ok(h, launch_best_alleles(best_params(*J.best, BL, W.dev, out_base, (uint32_t *)(W.dev + o_ri)), S_in), "phmm_best_alleles_kernel");
    if (good && PJ && !one_piece)  // the projection's inputs follow the head
        good = ok(h, hipMemcpyAsync(W.dev + o_pi, W.host + o_pi, p_end - o_pi, hipMemcpyHostToDevice, S_in), "H2D project");
    if (good && indexed && !one_piece) {
        memcpy(W.host + o_rb, J.ref_bases, rb);
        good = ok(h, hipMemcpyAsync(W.dev + o_rb, W.host + o_rb, rb, hipMemcpyHostToDevice, S_in), "H2D sw");
    }
    h->stat_staged_bytes += rb + ab;
    for (int c = 0; c < n_chunks && good; ++c) {
        const uint32_t a0 = cut[c], a1 = cut[c + 1];
        const size_t q0 = alt_off[a0], q1 = alt_off[a1];
        if (!one_piece && J.best && a1 > a0 && good) {
            // the likelihood rows of reads [a0, a1) are one range of the region-major matrix; then their best alleles
            const BestJob &B = *J.best;
            auto row_of = [&](uint32_t r, uint32_t *nh) -> uint64_t {
                const uint32_t g = (uint32_t)(std::upper_bound(B.region_read_off, B.region_read_off + B.n_regions + 1, r) - B.region_read_off) - 1;
                *nh = B.region_hap_off[g + 1] - B.region_hap_off[g];
                return B.out_off[g] + (uint64_t)(r - B.region_read_off[g]) * *nh;
            };
            uint32_t nh0 = 0, nh1 = 0;
            const uint64_t lo = row_of(a0, &nh0), hi = row_of(a1 - 1, &nh1) + nh1;
            if (hi > lo) {
                memcpy(W.host + BL.lk + 8ull * lo, B.likelihoods + lo, 8ull * (hi - lo));
                good = ok(h, hipMemcpyAsync(W.dev + BL.lk + 8ull * lo, W.host + BL.lk + 8ull * lo, 8ull * (hi - lo), hipMemcpyHostToDevice, S_in), "H2D sw");
            }
            BestParams bp = best_params(B, BL, W.dev, out_base, (uint32_t *)(W.dev + o_ri));
            bp.r_begin = a0;
            bp.n_reads = a1;
            good = good && ok(h, launch_best_alleles(bp, S_in), "phmm_best_alleles_kernel");
        }
        if (!one_piece) {
            if (!indexed) {  // one reference per alignment: they travel piece by piece like the alternates
                const size_t r0 = ref_off[a0], r1 = ref_off[a1];
                memcpy(W.host + o_rb + r0, J.ref_bases + r0, r1 - r0);
                good = r1 == r0 || ok(h, hipMemcpyAsync(W.dev + o_rb + r0, W.host + o_rb + r0, r1 - r0, hipMemcpyHostToDevice, S_in), "H2D sw");
            }
            memcpy(W.host + o_ab + q0, J.alt_bases + q0, q1 - q0);
            good = good && (q1 == q0 || ok(h, hipMemcpyAsync(W.dev + o_ab + q0, W.host + o_ab + q0, q1 - q0, hipMemcpyHostToDevice, S_in), "H2D sw")) &&
                   ok(h, hipEventRecord(W.ev_in[c], S_in), "hipEventRecord") && ok(h, hipStreamWaitEvent(S, W.ev_in[c], 0), "hipStreamWaitEvent");
        }
        if (!good || a1 == a0) continue;
        p.a_begin = a0;
        p.n_alignments = a1;
        const size_t workers = std::min<size_t>(max_workers, ((size_t)(a1 - a0) + gpb - 1) / gpb);
        (void)hipEventRecord(W.ev_k0[c], S);
        if (lite) {  // tags only, then the full instance over what met a gap (the counters are zeroed with the head of the input)
            SwParams p1 = p, p2 = p;
            p1.todo_out = (uint32_t *)(W.dev + o_todo) + (deferred || deferred_project ? 0 : a0);
            p1.todo_out_count = (uint32_t *)(W.dev + 192) + (deferred || deferred_project ? 0 : c);
            p2.todo = p1.todo_out;
            p2.todo_count = p1.todo_out_count;
            p2.feedback = zero_copy ? (uint32_t *)(W.host_dev + o_st + 192) + c : nullptr;  // (otherwise the counters come back with the status block)
            good = ok(h, launch_sw(L, K, transposed, SW_LITE, p1, (uint32_t)workers, lds, S), "phmm_sw_align_kernel (tags)");
            if (deferred || deferred_project) {  // (the second pass follows the last piece)
                (void)hipEventRecord(W.ev_k1[c], S);
                continue;
            }
            // The second pass of a large batch normally holds a handful of alignments, and in the batch's own geometry (eight to a
            // wave) even one costs a whole sweep (0.23 ms behind 2.8 ms): a list of up to 1 024 goes to the instance a small call
            // would get (one alignment per wave along the alternate, ~60 us), a longer one to the batch's; both launches look at
            // the counter and one of them returns at once.
            if (good && have_short) {
                SwParams ps = p2;
                ps.lds_ref_bytes = (uint32_t)GS.lds_ref;
                ps.lds_alt_bytes = (uint32_t)GS.lds_alt;
                ps.lds_group_bytes = (uint32_t)GS.lds_group;
                ps.groups_per_block = (uint32_t)GS.gpb;
                ps.slab_stride = GS.slab_stride;
                ps.todo_max = kShortList;
                p2.todo_min = kShortList + 1;
                const size_t fit = W.slab_bytes / (GS.slab_stride * 4);
                good = ok(h, launch_sw(GS.L, GS.K, GS.transposed, GS.variant, ps, (uint32_t)std::min<size_t>({(size_t)kShortList, fit, GS.max_workers}), GS.lds, S),
                          "phmm_sw_align_kernel (short list)");
            }
            good = good && ok(h, launch_sw(L, K, transposed, G.variant, p2, (uint32_t)workers, lds, S), "phmm_sw_align_kernel");
        } else {
            good = ok(h, launch_sw(L, K, transposed, G.variant, p, (uint32_t)workers, lds, S), "phmm_sw_align_kernel");
        }
        if (good && PJ) {  // ... and the piece's alignments projected onto the reference, where they lie
            pp.r_begin = a0;
            pp.n_reads = a1;
            good = ok(h, launch_project(pp, S), "phmm_project_kernel");
        }
        (void)hipEventRecord(W.ev_k1[c], S);
    }
    if ((deferred || deferred_project) && good) {
        if (!W.ev_second && !ok(h, hipEventCreate(&W.ev_second), "hipEventCreate")) return PHMM_ERR_HIP;
        SwParams p2 = p;
        p2.a_begin = 0;
        p2.n_alignments = n_alignments;
        p2.todo = (const uint32_t *)(W.dev + o_todo);
        p2.todo_count = (const uint32_t *)(W.dev + 192);
        if (have_short) {
            SwParams ps = p2;
            ps.lds_ref_bytes = (uint32_t)GS.lds_ref;
            ps.lds_alt_bytes = (uint32_t)GS.lds_alt;
            ps.lds_group_bytes = (uint32_t)GS.lds_group;
            ps.groups_per_block = (uint32_t)GS.gpb;
            ps.slab_stride = GS.slab_stride;
            ps.todo_max = kShortList;
            p2.todo_min = kShortList + 1;
            const size_t fit = W.slab_bytes / (GS.slab_stride * 4);
            good = ok(h, launch_sw(GS.L, GS.K, GS.transposed, GS.variant, ps, (uint32_t)std::min<size_t>({(size_t)kShortList, fit, GS.max_workers}), GS.lds, S),
                      "phmm_sw_align_kernel (short list)");
        }
        const size_t all_workers = std::min<size_t>({max_workers, ((size_t)n_alignments + gpb - 1) / gpb, W.slab_bytes / (slab_stride * 4)});
        good = good && ok(h, launch_sw(L, K, transposed, G.variant, p2, (uint32_t)all_workers, lds, S), "phmm_sw_align_kernel");
        if (deferred)
            good = good && ok(h, launch_sw_gather(p2.todo, p2.todo_count, p.n_cigar, p.alignment_offset, p.cigar, p.cigar_off, (uint32_t)max_slot,
                                                  kPatchMax, (uint32_t *)(W.dev + o_patch), S), "phmm_sw_gather_kernel");
        for (int c = 0; c < n_chunks && good && deferred_project; ++c) {  // (piece by piece: the workspace is a piece's)
            if (cut[c + 1] == cut[c]) continue;
            pp.r_begin = cut[c];
            pp.n_reads = cut[c + 1];
            good = ok(h, launch_project(pp, S), "phmm_project_kernel");
        }
        good = good && ok(h, hipEventRecord(W.ev_second, S), "hipEventRecord");
    }
    // (while the device works) what the kernels store per alignment: (rows + L - 1) steps x L lanes x flag words per strip
    W.last_backtrack_bytes = 0;
    for (uint32_t a = 0; a < n_alignments; ++a) {
        const uint32_t ri = J.ref_index ? J.ref_index[a] : a;
        if (ri == SW_NO_REFERENCE) continue;
        const uint64_t rows = J.best ? max_ref : ref_off[ri + 1] - ref_off[ri];  // (the device chooses the haplotype: an upper bound)
        const uint64_t cols = alt_off[a + 1] - alt_off[a];
        const uint64_t words = lite ? 0ull : flag_words;  // (the tags-only sweep stores none since round 5; what the second pass adds for the
                                                          // alignments with gaps is not counted)
        W.last_backtrack_bytes += transposed ? (cols + L - 1ull) * L * words * 4ull
                                             : (uint64_t)((cols + strip_cols - 1) / strip_cols) * (rows + L - 1ull) * L * words * 4ull;
    }
    // Results come back piece by piece, on a stream of their own: a piece's D2H is issued once the host has seen
    // its kernel finish (a copy that waits in the queue for a kernel holds back the H2D copies behind it, NOTEBOOK.md
    // section 9), and is unpacked into the caller's arrays while the later pieces compute.
    hipStream_t S_out = one_piece ? S : h->streams[2];
    auto unpack = [&](int c) {
        const uint32_t a0 = cut[c], a1 = cut[c + 1];
        if (!one_piece && !ok(h, hipEventSynchronize(W.ev_out[c]), "sync(sw results)")) return false;  // (one piece: the stream has been waited for)
        if (J.view && !PJ) return true;
        if (PJ) {
            memcpy(PJ->status + a0, W.host + o_pst + 4ull * a0, 4ull * (a1 - a0));
            memcpy(PJ->n_out_cigar + a0, W.host + o_pno + 4ull * a0, 4ull * (a1 - a0));
            memcpy(PJ->new_pos + a0, W.host + o_ppos + 8ull * a0, 8ull * (a1 - a0));
            const uint64_t q0 = PJ->out_cigar_off[a0], q1 = PJ->out_cigar_off[a1];
            if (q1 > q0) memcpy(PJ->out_cigar + q0, W.host + o_pout + 4ull * q0, 4ull * (q1 - q0));
            return true;
        }
        memcpy(J.n_cigar + a0, W.host + o_nc + 4ull * a0, 4ull * (a1 - a0));
        memcpy(J.alignment_offset + a0, W.host + o_of + 4ull * a0, 4ull * (a1 - a0));
        if (cigar_off[a1] > cigar_off[a0])
            memcpy(J.cigar + cigar_off[a0], W.host + o_cg + 4ull * cigar_off[a0], 4ull * (cigar_off[a1] - cigar_off[a0]));
        return true;
    };
    int prev = -1, last_piece = 0;
    for (int c = 0; c < n_chunks; ++c)
        if (cut[c + 1] > cut[c]) last_piece = c;
    bool best_fetched = false;
    for (int c = 0; c < n_chunks && good; ++c) {
        const uint32_t a0 = cut[c], a1 = cut[c + 1];
        if (a1 == a0) continue;
        const uint64_t g0 = cigar_off[a0], g1 = cigar_off[a1];
        good = one_piece || ok(h, hipEventSynchronize(deferred_project ? W.ev_second : W.ev_k1[c]), "sync(sw kernel)");
        if (zero_copy) {
            // nothing to fetch
        } else if (one_piece) {  // the piece is the call: its results are one contiguous block of the staging buffer
            const size_t from = PJ ? o_pst : o_nc, to = PJ ? o_extra : o_pfl;
            good = good && (J.view && !PJ ? true : ok(h, hipMemcpyAsync(W.host + from, W.dev + from, to - from, hipMemcpyDeviceToHost, S_out), "D2H sw"));
        } else if (PJ) {
            const uint64_t q0 = PJ->out_cigar_off[a0], q1 = PJ->out_cigar_off[a1];
            good = good &&
                   ok(h, hipMemcpyAsync(W.host + o_pst + 4ull * a0, W.dev + o_pst + 4ull * a0, 4ull * (a1 - a0), hipMemcpyDeviceToHost, S_out), "D2H project") &&
                   ok(h, hipMemcpyAsync(W.host + o_pno + 4ull * a0, W.dev + o_pno + 4ull * a0, 4ull * (a1 - a0), hipMemcpyDeviceToHost, S_out), "D2H project") &&
                   ok(h, hipMemcpyAsync(W.host + o_ppos + 8ull * a0, W.dev + o_ppos + 8ull * a0, 8ull * (a1 - a0), hipMemcpyDeviceToHost, S_out), "D2H project") &&
                   (q1 == q0 || ok(h, hipMemcpyAsync(W.host + o_pout + 4ull * q0, W.dev + o_pout + 4ull * q0, 4ull * (q1 - q0), hipMemcpyDeviceToHost, S_out), "D2H project"));
        } else if (!J.view) {
            good = good &&
                   ok(h, hipMemcpyAsync(W.host + o_nc + 4ull * a0, W.dev + o_nc + 4ull * a0, 4ull * (a1 - a0), hipMemcpyDeviceToHost, S_out), "D2H sw") &&
                   ok(h, hipMemcpyAsync(W.host + o_of + 4ull * a0, W.dev + o_of + 4ull * a0, 4ull * (a1 - a0), hipMemcpyDeviceToHost, S_out), "D2H sw") &&
                   (g1 == g0 || ok(h, hipMemcpyAsync(W.host + o_cg + 4ull * g0, W.dev + o_cg + 4ull * g0, 4ull * (g1 - g0), hipMemcpyDeviceToHost, S_out), "D2H sw"));
        }
        good = good && (one_piece || ok(h, hipEventRecord(W.ev_out[c], S_out), "hipEventRecord"));
        if (good && J.best && !best_fetched && !zero_copy && c == last_piece) {  // (final once the last piece's kernel has been seen to end)
            good = ok(h, hipMemcpyAsync(W.host + BL.best, W.dev + BL.best, BL.end - BL.best, hipMemcpyDeviceToHost, S_out), "D2H best alleles");
            best_fetched = true;
        }
        if (good && prev >= 0) good = unpack(prev);
        prev = c;
    }
    if (deferred && good)  // the status block and the gathered alignments are final once the second pass is through
        good = ok(h, hipEventSynchronize(W.ev_second), "sync(sw second pass)") &&
               ok(h, hipMemcpyAsync(W.host + o_patch, W.dev + o_patch, patch_bytes, hipMemcpyDeviceToHost, S_out), "D2H sw");
    good = good && (zero_copy || ok(h, hipMemcpyAsync(W.host + o_st, W.dev, 256, hipMemcpyDeviceToHost, S_out), "D2H sw"));
    if (one_piece) {
        good = good && ok(h, hipStreamSynchronize(S_out), "sync(sw)");
        if (good && prev >= 0) good = unpack(prev);
    } else {
        if (good && prev >= 0) good = unpack(prev);
        good = good && ok(h, hipStreamSynchronize(S_out), "sync(sw)");
    }
    if (!good) {
        (void)hipStreamSynchronize(S_in);
        (void)hipStreamSynchronize(S);
        (void)hipStreamSynchronize(S_out);
        return PHMM_ERR_HIP;
    }
    if (deferred) {  // what the second pass redid, over what the pieces brought
        const uint32_t again = *(const uint32_t *)(W.host + o_st + 192);
        if (again <= kPatchMax) {
            const uint32_t *e = (const uint32_t *)(W.host + o_patch);
            for (uint32_t t = 0; t < again; ++t, e += 3 + max_slot) {
                const uint32_t a = e[0];
                J.n_cigar[a] = e[1];
                J.alignment_offset[a] = (int32_t)e[2];
                const uint64_t slot = cigar_off[a + 1] - cigar_off[a];
                if (slot) memcpy(J.cigar + cigar_off[a], e + 3, 4ull * std::min<uint64_t>(slot, e[1]));
            }
        } else {  // (a call full of gaps: everything once more; the handle then stops taking the first pass)
            if (!ok(h, hipMemcpy(W.host + o_nc, W.dev + o_nc, o_pfl - o_nc, hipMemcpyDeviceToHost), "D2H sw")) return PHMM_ERR_HIP;
            memcpy(J.n_cigar, W.host + o_nc, 4ull * n_alignments);
            memcpy(J.alignment_offset, W.host + o_of, 4ull * n_alignments);
            if (n_cig) memcpy(J.cigar, W.host + o_cg, 4ull * n_cig);
        }
    }
    if (J.best) {
        memcpy(J.best->best_allele, W.host + BL.best, 4ull * J.best->n_reads);
        memcpy(J.best->likelihood, W.host + BL.olk, 8ull * J.best->n_reads);
        memcpy(J.best->confidence, W.host + BL.conf, 8ull * J.best->n_reads);
    }
    W.last_kernel_us = 0;
    for (int c = 0; c < n_chunks; ++c) {
        float ms = 0.f;
        if (cut[c + 1] > cut[c] && hipEventElapsedTime(&ms, W.ev_k0[c], W.ev_k1[c]) == hipSuccess) W.last_kernel_us += (uint64_t)(ms * 1e3f);
    }
    if (deferred || deferred_project) {  // ... and the second pass behind the last piece
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, W.ev_k1[last_piece], W.ev_second) == hipSuccess) W.last_kernel_us += (uint64_t)(ms * 1e3f);
    }
    if (lite) {
        uint64_t again = 0;
        for (int c = 0; c < (deferred || deferred_project ? 1 : n_chunks); ++c) again += ((const uint32_t *)(W.host + o_st + 192))[c];
        W.last_second_pass = again;
        if (again * 10 > (uint64_t)n_alignments * 3) W.lite_skip = 15;
    } else {
        W.last_second_pass = 0;
    }
    const uint32_t *st = (const uint32_t *)(W.host + o_st + 64);  // [0], [1] conditions; [2], [3]: shader clocks / 100 MHz ticks of the last kernel's block 0
    W.last_clock_mhz = st[3] ? (uint64_t)((double)st[2] * 100.0 / (double)st[3]) : 0;
    if (st[SW_STATUS_EMPTY]) return fail(h, who + empty_msg);  // (an alignment the device met had an empty sequence)
    if (st[SW_STATUS_CAPACITY]) {
        if (on_device) {  // an alignment outgrew the library's own slots: tell the caller how large the largest is (it runs again)
            std::vector<uint32_t> n_cig_host(n_alignments);
            if (!ok(h, hipMemcpy(n_cig_host.data(), W.dev + o_nc, 4ull * n_alignments, hipMemcpyDeviceToHost), "D2H sw")) return PHMM_ERR_HIP;
            if (J.sw_capacity_needed) *J.sw_capacity_needed = *std::max_element(n_cig_host.begin(), n_cig_host.end());
        }
        h->err = who + ": a CIGAR needs more elements than its slot holds (n_cigar has the sizes)";
        return h->err_code = PHMM_ERR_CIGAR_CAPACITY;
    }
    if (PJ && (*(const uint32_t *)(W.host + o_st + 128) & 1u)) {
        h->err = who + ": a CIGAR needs more elements than its slot holds (n_out_cigar has the sizes)";
        return h->err_code = PHMM_ERR_CIGAR_CAPACITY;
    }
    if (J.view) {
        J.view->ref_off = p.ref_off;
        J.view->alt_off = p.alt_off;
        J.view->ref_bases = p.ref_bases;
        J.view->alt_bases = p.alt_bases;
        J.view->cigar_off = p.cigar_off;
        J.view->cigar = p.cigar;
        J.view->n_cigar = p.n_cigar;
        J.view->alignment_offset = p.alignment_offset;
        J.view->extra_dev = W.dev + o_extra;
        J.view->extra_host = W.host + o_extra;
    }
    return PHMM_OK;
}

template <class F>
int guarded(phmm_handle *h, const char *who, F &&body) {
    if (!h) return PHMM_ERR_INVALID_ARG;
    try {
        phmm_host::latch_slot0(h);
        return body();
    } catch (const std::bad_alloc &) {
        h->err = std::string(who) + ": out of host memory";
        return h->err_code = PHMM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        h->err = std::string(who) + ": " + e.what();
        return h->err_code = PHMM_ERR_INTERNAL;
    }
}

}  // namespace

extern "C" int phmm_sw_align(phmm_handle *h, uint32_t n_alignments, const uint32_t *ref_off, const uint8_t *ref_bases,
                             const uint32_t *alt_off, const uint8_t *alt_bases, const phmm_sw_parameters *params,
                             int overhang_strategy, const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar,
                             int32_t *alignment_offset) {
    return guarded(h, "phmm_sw_align", [&]() -> int {
        SwJob J;
        J.n_alignments = J.n_refs = n_alignments;
        J.ref_off = ref_off;
        J.ref_bases = ref_bases;
        J.alt_off = alt_off;
        J.alt_bases = alt_bases;
        J.params = params;
        J.strategy = overhang_strategy;
        J.cigar_off = cigar_off;
        J.cigar = cigar;
        J.n_cigar = n_cigar;
        J.alignment_offset = alignment_offset;
        return sw_run(h, J);
    });
}

extern "C" int phmm_sw_align_indexed(phmm_handle *h, uint32_t n_references, const uint32_t *ref_off, const uint8_t *ref_bases,
                                     uint32_t n_alignments, const uint32_t *ref_index, const uint32_t *alt_off,
                                     const uint8_t *alt_bases, const phmm_sw_parameters *params, int overhang_strategy,
                                     const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar, int32_t *alignment_offset) {
    return guarded(h, "phmm_sw_align_indexed", [&]() -> int {
        if (n_alignments && !ref_index) return fail(h, "phmm_sw_align_indexed: null array");
        SwJob J;
        J.who = "phmm_sw_align_indexed";
        J.n_alignments = n_alignments;
        J.n_refs = n_references;
        J.ref_off = ref_off;
        J.ref_bases = ref_bases;
        J.ref_index = ref_index;
        J.alt_off = alt_off;
        J.alt_bases = alt_bases;
        J.params = params;
        J.strategy = overhang_strategy;
        J.cigar_off = cigar_off;
        J.cigar = cigar;
        J.n_cigar = n_cigar;
        J.alignment_offset = alignment_offset;
        return sw_run(h, J);
    });
}

extern "C" int phmm_best_alleles(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off,
                                 const uint32_t *region_hap_off, const uint64_t *out_off, const double *likelihoods,
                                 const uint8_t *keep, const int32_t *hap_priority, double informative_threshold,
                                 int32_t *best_allele, double *likelihood, double *confidence) {
    return guarded(h, "phmm_best_alleles", [&]() -> int {
        h->err_code = PHMM_OK;
        BestJob B;
        B.n_regions = n_regions;
        B.region_read_off = region_read_off;
        B.region_hap_off = region_hap_off;
        B.out_off = out_off;
        B.likelihoods = likelihoods;
        B.keep = keep;
        B.priority = hap_priority;
        B.threshold = informative_threshold;
        B.best_allele = best_allele;
        B.likelihood = likelihood;
        B.confidence = confidence;
        if (n_regions && region_read_off && region_hap_off) {
            B.n_reads = region_read_off[n_regions];
            B.n_haps = region_hap_off[n_regions];
        }
        const int st = check_best(h, "phmm_best_alleles", B);
        if (st != PHMM_OK || !n_regions || !B.n_reads) return st;
        DevGuard dg(h->device);
        const BestLayout BL(&B, 0);
        if (!grow_staging(h, BL.end)) return PHMM_ERR_HIP;
        phmm_handle::SwWork &W = h->swork;
        hipStream_t S = h->streams[0];
        stage_best(B, BL, W.host);
        if (!ok(h, hipMemcpyAsync(W.dev, W.host, BL.best, hipMemcpyHostToDevice, S), "H2D best alleles") ||
            !ok(h, launch_best_alleles(best_params(B, BL, W.dev, W.dev, nullptr), S), "phmm_best_alleles_kernel") ||
            !ok(h, hipMemcpyAsync(W.host + BL.best, W.dev + BL.best, BL.end - BL.best, hipMemcpyDeviceToHost, S), "D2H best alleles") ||
            !ok(h, hipStreamSynchronize(S), "sync(best alleles)"))
            return PHMM_ERR_HIP;
        memcpy(best_allele, W.host + BL.best, 4ull * B.n_reads);
        memcpy(likelihood, W.host + BL.olk, 8ull * B.n_reads);
        memcpy(confidence, W.host + BL.conf, 8ull * B.n_reads);
        return PHMM_OK;
    });
}

extern "C" int phmm_realign_to_best(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off,
                                    const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases,
                                    const uint32_t *hap_off, const uint8_t *hap_bases, const uint64_t *out_off,
                                    const double *likelihoods, const uint8_t *keep, const int32_t *hap_priority,
                                    double informative_threshold, const phmm_sw_parameters *params, int overhang_strategy,
                                    const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar, int32_t *alignment_offset,
                                    int32_t *best_allele, double *likelihood, double *confidence) {
    return guarded(h, "phmm_realign_to_best", [&]() -> int {
        if (n_regions && (!region_read_off || !region_hap_off)) return fail(h, "phmm_realign_to_best: null array");
        BestJob B;
        B.n_regions = n_regions;
        B.region_read_off = region_read_off;
        B.region_hap_off = region_hap_off;
        B.out_off = out_off;
        B.likelihoods = likelihoods;
        B.keep = keep;
        B.priority = hap_priority;
        B.threshold = informative_threshold;
        B.best_allele = best_allele;
        B.likelihood = likelihood;
        B.confidence = confidence;
        B.n_reads = n_regions ? region_read_off[n_regions] : 0;
        B.n_haps = n_regions ? region_hap_off[n_regions] : 0;
        SwJob J;
        J.who = "phmm_realign_to_best";
        J.n_alignments = B.n_reads;
        J.n_refs = B.n_haps;
        J.ref_off = hap_off;
        J.ref_bases = hap_bases;
        J.alt_off = read_off;
        J.alt_bases = read_bases;
        J.params = params;
        J.strategy = overhang_strategy;
        J.cigar_off = cigar_off;
        J.cigar = cigar;
        J.n_cigar = n_cigar;
        J.alignment_offset = alignment_offset;
        J.best = &B;
        if (B.n_reads && !B.n_haps) {  // no alleles at all: search_best_allele's None for every read (:465-475), nothing to align
            h->err_code = PHMM_OK;
            const int st = check_best(h, J.who, B);
            if (st != PHMM_OK) return st;
            if (!n_cigar || !alignment_offset) return fail(h, "phmm_realign_to_best: null array");
            for (uint32_t r = 0; r < B.n_reads; ++r) {
                best_allele[r] = -1;
                likelihood[r] = -HUGE_VAL;
                confidence[r] = std::nan("");
                n_cigar[r] = 0;
                alignment_offset[r] = 0;
            }
            return PHMM_OK;
        }
        return sw_run(h, J);
    });
}

extern "C" int phmm_realign_reads(phmm_handle *h, uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                                  const uint32_t *read_off, const uint8_t *read_bases, const uint32_t *hap_off,
                                  const uint8_t *hap_bases, const uint64_t *out_off, const double *likelihoods, const uint8_t *keep,
                                  const int32_t *hap_priority, double informative_threshold, const phmm_sw_parameters *params,
                                  int overhang_strategy, const int32_t *region_ref_hap, const uint64_t *region_reference_start,
                                  const uint32_t *hap_cigar_off, const uint32_t *hap_cigar, const uint32_t *hap_start_wrt_ref,
                                  const uint32_t *orig_cigar_off, const uint32_t *orig_cigar, const uint64_t *out_cigar_off,
                                  uint32_t *out_cigar, uint32_t *n_out_cigar, int64_t *new_pos, int32_t *status,
                                  int32_t *best_allele, double *likelihood, double *confidence) {
    return guarded(h, "phmm_realign_reads", [&]() -> int {
        if (n_regions && (!region_read_off || !region_hap_off)) return fail(h, "phmm_realign_reads: null array");
        BestJob B;
        B.n_regions = n_regions;
        B.region_read_off = region_read_off;
        B.region_hap_off = region_hap_off;
        B.out_off = out_off;
        B.likelihoods = likelihoods;
        B.keep = keep;
        B.priority = hap_priority;
        B.threshold = informative_threshold;
        B.best_allele = best_allele;
        B.likelihood = likelihood;
        B.confidence = confidence;
        B.n_reads = n_regions ? region_read_off[n_regions] : 0;
        B.n_haps = n_regions ? region_hap_off[n_regions] : 0;
        ProjJob P;
        P.region_ref_hap = region_ref_hap;
        P.region_reference_start = region_reference_start;
        P.hap_cigar_off = hap_cigar_off;
        P.hap_cigar = hap_cigar;
        P.hap_start_wrt_ref = hap_start_wrt_ref;
        P.orig_cigar_off = orig_cigar_off;
        P.orig_cigar = orig_cigar;
        P.out_cigar_off = out_cigar_off;
        P.out_cigar = out_cigar;
        P.n_out_cigar = n_out_cigar;
        P.new_pos = new_pos;
        P.status = status;
        if (hap_cigar_off)
            for (uint32_t a = 0; a < B.n_haps; ++a)
                if (hap_cigar_off[a + 1] >= hap_cigar_off[a]) P.max_hap_cigar = std::max(P.max_hap_cigar, hap_cigar_off[a + 1] - hap_cigar_off[a]);
        SwJob J;
        J.who = "phmm_realign_reads";
        J.n_alignments = B.n_reads;
        J.n_refs = B.n_haps;
        J.ref_off = hap_off;
        J.ref_bases = hap_bases;
        J.alt_off = read_off;
        J.alt_bases = read_bases;
        J.params = params;
        J.strategy = overhang_strategy;
        J.best = &B;
        J.proj = &P;
        uint32_t needed = 0;
        J.sw_capacity_needed = &needed;
        if (B.n_reads && !B.n_haps) {  // no alleles at all: nothing to align, every read stays as it is
            h->err_code = PHMM_OK;
            const int st = check_best(h, J.who, B);
            if (st != PHMM_OK) return st;
            if (!n_out_cigar || !new_pos || !status) return fail(h, "phmm_realign_reads: null array");
            for (uint32_t r = 0; r < B.n_reads; ++r) {
                best_allele[r] = -1;
                likelihood[r] = -HUGE_VAL;
                confidence[r] = std::nan("");
                n_out_cigar[r] = 0;
                new_pos[r] = 0;
                status[r] = CIGAR_UNCHANGED;
            }
            return PHMM_OK;
        }
        int st = sw_run(h, J);
        if (st == PHMM_ERR_CIGAR_CAPACITY && needed > P.sw_capacity) {  // once more with slots as large as the largest alignment needs
            P.sw_capacity = needed;
            needed = 0;
            st = sw_run(h, J);
        }
        return st;
    });
}

extern "C" int phmm_calculate_cigar(phmm_handle *h, uint32_t n, const uint32_t *ref_off, const uint8_t *ref_bases, const uint32_t *alt_off,
                                    const uint8_t *alt_bases, const phmm_sw_parameters *params, int overhang_strategy,
                                    const uint64_t *cigar_off, uint32_t *cigar, uint32_t *n_cigar, int32_t *status) {
    return guarded(h, "phmm_calculate_cigar", [&]() -> int {
        h->err_code = PHMM_OK;
        if (!n) return PHMM_OK;
        if (!ref_off || !alt_off || !cigar_off || !n_cigar || !status) return fail(h, "phmm_calculate_cigar: null array");
        if (ref_off[0] != 0 || alt_off[0] != 0 || cigar_off[0] != 0) return fail(h, "phmm_calculate_cigar: offset arrays must start at 0");
        for (uint32_t a = 0; a < n; ++a)
            if (ref_off[a + 1] < ref_off[a] || alt_off[a + 1] < alt_off[a] || cigar_off[a + 1] < cigar_off[a])
                return fail(h, "phmm_calculate_cigar: offsets not monotonic");
        if ((ref_off[n] && !ref_bases) || (alt_off[n] && !alt_bases) || (cigar_off[n] && !cigar)) return fail(h, "phmm_calculate_cigar: null array");
        if ((uint64_t)ref_off[n] + 2ull * SW_PAD_BASES * n > 0xffffffffull || (uint64_t)alt_off[n] + 2ull * SW_PAD_BASES * n > 0xffffffffull)
            return fail(h, "phmm_calculate_cigar: too many bases for one call (4 GB with the padding)");
        // both sequences of every pair between two runs of SW_PAD (cigar_utils.rs:387-398)
        std::vector<uint32_t> p_ref_off(n + 1), p_alt_off(n + 1);
        for (uint32_t a = 0; a <= n; ++a) {
            p_ref_off[a] = ref_off[a] + 2 * SW_PAD_BASES * a;
            p_alt_off[a] = alt_off[a] + 2 * SW_PAD_BASES * a;
        }
        std::vector<uint8_t> p_ref(p_ref_off[n], (uint8_t)'N'), p_alt(p_alt_off[n], (uint8_t)'N');
        for (uint32_t a = 0; a < n; ++a) {
            if (ref_off[a + 1] > ref_off[a]) memcpy(p_ref.data() + p_ref_off[a] + SW_PAD_BASES, ref_bases + ref_off[a], ref_off[a + 1] - ref_off[a]);
            if (alt_off[a + 1] > alt_off[a]) memcpy(p_alt.data() + p_alt_off[a] + SW_PAD_BASES, alt_bases + alt_off[a], alt_off[a + 1] - alt_off[a]);
        }
        const uint64_t n_out = cigar_off[n];
        SwDeviceView V;
        const size_t x_fl = 0, x_st = 256, x_no = x_st + up256(4ull * n), x_oo = x_no + up256(4ull * n), x_out = x_oo + up256(8ull * (n + 1));
        V.extra_bytes = x_out + up256(4ull * n_out);
        SwJob J;
        J.who = "phmm_calculate_cigar";
        J.n_alignments = J.n_refs = n;
        J.ref_off = p_ref_off.data();
        J.ref_bases = p_ref.data();
        J.alt_off = p_alt_off.data();
        J.alt_bases = p_alt.data();
        J.params = params;
        J.strategy = overhang_strategy;
        J.view = &V;
        uint32_t needed = 0;
        J.sw_capacity_needed = &needed;
        int st = sw_run(h, J);
        if (st == PHMM_ERR_CIGAR_CAPACITY && needed > V.sw_capacity) {  // once more with slots as large as the largest alignment needs
            V.sw_capacity = needed;
            st = sw_run(h, J);
        }
        if (st != PHMM_OK) return st;
        DevGuard dg(h->device);
        phmm_handle::SwWork &W = h->swork;
        hipStream_t S = h->streams[0];
        // three arrays per lane: the trimmed cigar (+ one element), left_align_indels' right-to-left list (four per element + 2), its result
        const uint32_t capacity = 4 * (V.sw_capacity + 2) + 8;
        const size_t ws_bytes = (size_t)n * 3 * capacity * 4;
        if (W.ws_bytes < ws_bytes) {
            (void)hipStreamSynchronize(S);
            if (W.ws) (void)hipFree(W.ws);
            W.ws = nullptr;
            W.ws_bytes = 0;
            if (!ok(h, hipMalloc((void **)&W.ws, ws_bytes), "hipMalloc(cigar workspace)")) return PHMM_ERR_HIP;
            W.ws_bytes = ws_bytes;
        }
        memset(V.extra_host + x_fl, 0, 256);
        memcpy(V.extra_host + x_oo, cigar_off, 8ull * (n + 1));
        CalcParams c{};
        c.n = n;
        c.ref_off = V.ref_off;
        c.alt_off = V.alt_off;
        c.ref_bases = V.ref_bases;
        c.alt_bases = V.alt_bases;
        c.sw_cigar_off = V.cigar_off;
        c.sw_cigar = V.cigar;
        c.n_sw_cigar = V.n_cigar;
        c.sw_offset = V.alignment_offset;
        c.out_cigar_off = (const uint64_t *)(V.extra_dev + x_oo);
        c.out_cigar = (uint32_t *)(V.extra_dev + x_out);
        c.n_out_cigar = (uint32_t *)(V.extra_dev + x_no);
        c.status = (int32_t *)(V.extra_dev + x_st);
        c.flags = (uint32_t *)(V.extra_dev + x_fl);
        c.workspace = W.ws;
        c.capacity = capacity;
        if (!ok(h, hipMemcpyAsync(V.extra_dev + x_fl, V.extra_host + x_fl, 256, hipMemcpyHostToDevice, S), "H2D cigar") ||
            !ok(h, hipMemcpyAsync(V.extra_dev + x_oo, V.extra_host + x_oo, 8ull * (n + 1), hipMemcpyHostToDevice, S), "H2D cigar") ||
            !ok(h, launch_calculate_cigar(c, S), "phmm_calculate_cigar_kernel") ||
            !ok(h, hipMemcpyAsync(V.extra_host, V.extra_dev, x_oo, hipMemcpyDeviceToHost, S), "D2H cigar") ||
            (n_out && !ok(h, hipMemcpyAsync(V.extra_host + x_out, V.extra_dev + x_out, 4ull * n_out, hipMemcpyDeviceToHost, S), "D2H cigar")) ||
            !ok(h, hipStreamSynchronize(S), "sync(cigar)"))
            return PHMM_ERR_HIP;
        memcpy(status, V.extra_host + x_st, 4ull * n);
        memcpy(n_cigar, V.extra_host + x_no, 4ull * n);
        if (n_out) memcpy(cigar, V.extra_host + x_out, 4ull * n_out);
        if (*(const uint32_t *)(V.extra_host + x_fl) & 1u) {
            h->err = "phmm_calculate_cigar: a CIGAR needs more elements than its slot holds (n_cigar has the sizes)";
            return h->err_code = PHMM_ERR_CIGAR_CAPACITY;
        }
        return PHMM_OK;
    });
}
