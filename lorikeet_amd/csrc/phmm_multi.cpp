// phmm_assign_regions / phmm_split_regions / phmm_compute_multi (include/phmm.h): one call over several engines, one per
// device, from one process -- what a single Lorikeet process on a multi-GPU node needs.  Whole regions go to
// engines in contiguous cell-balanced ranges (or, for heavy-tailed sets, one by one by greedy longest-processing-time on
// cells(region)); every engine computes its share concurrently, on a host thread of its own pinned next to its GPU,
// staging straight from the caller's arrays into its pinned mirror and writing into disjoint slices of `out`.  Nothing
// is gathered first ("expected limiter is host-side marshalling"), there is no exchange between devices and
// no CPU fallback.
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "phmm_host.hpp"

using namespace phmm_host;

namespace {

// cells(region) = sum of read lengths x sum of haplotype lengths (the metric's unit)
std::vector<uint64_t> region_cells(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                                   const uint32_t *read_off, const uint32_t *hap_off) {
    std::vector<uint64_t> cells(n_regions);
    for (uint32_t g = 0; g < n_regions; ++g) {
        const uint64_t sr = (uint64_t)read_off[region_read_off[g + 1]] - read_off[region_read_off[g]];
        const uint64_t sh = (uint64_t)hap_off[region_hap_off[g + 1]] - hap_off[region_hap_off[g]];
        cells[g] = sr * sh;
    }
    return cells;
}

// Heaviest region first onto the least loaded part; ties by part index, then region index (the order the Python
// sharding of the multi-process path uses, so both give the same assignment).
void assign_lpt(const std::vector<uint64_t> &cells, uint32_t n_parts, uint32_t *part_of_region) {
    std::vector<uint32_t> order(cells.size());
    for (uint32_t g = 0; g < order.size(); ++g) order[g] = g;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cells[a] > cells[b]; });
    typedef std::pair<uint64_t, uint32_t> Load;  // (cells so far, part)
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (uint32_t p = 0; p < n_parts; ++p) heap.push({0, p});
    for (uint32_t g : order) {
        Load l = heap.top();
        heap.pop();
        part_of_region[g] = l.second;
        heap.push({l.first + cells[g], l.second});
    }
}

// Boundaries of n_parts contiguous, cell-balanced ranges: boundary k is the prefix position closest to k/n_parts of the
// total (ties to the left), kept monotone -- the rule of lorikeet_amd/sharding.py:split_contiguous.
void split_contiguous(const std::vector<uint64_t> &cells, uint32_t n_parts, uint32_t *first_region) {
    const uint32_t n = (uint32_t)cells.size();
    std::vector<unsigned __int128> prefix(n + 1, 0);
    for (uint32_t g = 0; g < n; ++g) prefix[g + 1] = prefix[g] + cells[g];
    const unsigned __int128 total = prefix[n];
    first_region[0] = 0;
    uint32_t g = 0;
    for (uint32_t k = 1; k < n_parts; ++k) {
        while (g < n && prefix[g] * n_parts < total * k) ++g;
        if (g > first_region[k - 1] && g > 0 && (total * k - prefix[g - 1] * n_parts) <= (prefix[g] * n_parts - total * k)) --g;
        g = std::max(g, first_region[k - 1]);
        first_region[k] = g;
    }
    first_region[n_parts] = n;
}

// The calling thread moves next to its engine's GPU: the CPUs the device's PCI function lists as local
// (/sys/bus/pci/devices/<bdf>/local_cpulist), intersected with what the thread may use.  Staging copies then run out
// of the memory of the GPU's own NUMA node.  Best effort: any failure leaves the affinity alone.
void pin_near_device(int device) {
    char bdf[32] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return;
    for (char *p = bdf; *p; ++p) *p = (char)tolower(*p);
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return;
    cpu_set_t allowed, want;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    CPU_ZERO(&want);
    int n_want = 0;
    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int lo = 0, hi = 0;
        const int k = sscanf(tok, "%d-%d", &lo, &hi);
        if (k < 1) continue;
        if (k == 1) hi = lo;
        for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &allowed)) {
                CPU_SET(c, &want);
                ++n_want;
            }
    }
    if (n_want > 0) (void)sched_setaffinity(0, sizeof want, &want);
}

}  // namespace

extern "C" {

int phmm_assign_regions(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                        const uint32_t *read_off, const uint32_t *hap_off, uint32_t n_parts, uint32_t *part_of_region) {
    if (!n_parts || (n_regions && (!region_read_off || !region_hap_off || !read_off || !hap_off || !part_of_region)))
        return PHMM_ERR_INVALID_ARG;
    try {
        assign_lpt(region_cells(n_regions, region_read_off, region_hap_off, read_off, hap_off), n_parts, part_of_region);
    } catch (...) {
        return PHMM_ERR_NO_MEMORY;
    }
    return PHMM_OK;
}

int phmm_split_regions(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                       const uint32_t *read_off, const uint32_t *hap_off, uint32_t n_parts, uint32_t *first_region) {
    if (!n_parts || !first_region || (n_regions && (!region_read_off || !region_hap_off || !read_off || !hap_off)))
        return PHMM_ERR_INVALID_ARG;
    try {
        split_contiguous(region_cells(n_regions, region_read_off, region_hap_off, read_off, hap_off), n_parts, first_region);
    } catch (...) {
        return PHMM_ERR_NO_MEMORY;
    }
    return PHMM_OK;
}

int phmm_compute_multi(phmm_handle *const *handles, uint32_t n_handles, uint32_t n_regions, const uint32_t *region_read_off,
                       const uint32_t *region_hap_off, const uint32_t *read_off, const uint8_t *read_bases,
                       const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *gcp,
                       const uint32_t *hap_off, const uint8_t *hap_bases, const uint64_t *out_off, double *out) {
    if (!handles || !n_handles || !handles[0]) return PHMM_ERR_INVALID_ARG;
    phmm_handle *h0 = handles[0];  // carries the message of a failed call (phmm_last_error(handles[0]))
    for (uint32_t k = 0; k < n_handles; ++k)
        if (!handles[k]) {
            h0->err = "phmm_compute_multi: null handle";
            return PHMM_ERR_INVALID_ARG;
        }
    clear_thread_error(h0);
    if (const char *bad = validate_offsets(n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, nullptr)) {
        h0->err = bad;
        return PHMM_ERR_INVALID_ARG;
    }
    if (n_handles == 1)
        return phmm_compute(h0, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, gcp,
                            hap_off, hap_bases, out_off, out);
    const uint32_t n_reads = region_read_off[n_regions], n_haps = region_hap_off[n_regions];
    if ((read_off[n_reads] && (!read_bases || !base_q || !ins_q || !del_q || !gcp)) || (hap_off[n_haps] && !hap_bases) ||
        (out_off[n_regions] && !out)) {
        h0->err = "phmm_compute_multi: null pointer";
        return PHMM_ERR_INVALID_ARG;
    }
    try {
        // Contiguous, cell-balanced ranges need no gather at all: every engine stages its range straight from the
        // caller's arrays.  Only when one heavy region makes such ranges uneven (a part more than 5 % above the mean)
        // are regions dealt out one by one (greedy LPT) -- still without a gather: every engine walks its own list
        // and stages region by region from the caller's arrays.
        const std::vector<uint64_t> cells = region_cells(n_regions, region_read_off, region_hap_off, read_off, hap_off);
        std::vector<uint32_t> first(n_handles + 1);
        split_contiguous(cells, n_handles, first.data());
        unsigned __int128 total = 0;
        uint64_t heaviest = 0;
        for (uint32_t k = 0; k < n_handles; ++k) {
            uint64_t load = 0;
            for (uint32_t g = first[k]; g < first[k + 1]; ++g) load += cells[g];
            heaviest = std::max(heaviest, load);
            total += load;
        }
        const bool contiguous = (unsigned __int128)heaviest * n_handles * 100 <= total * 105;
        std::vector<std::vector<uint32_t>> lists;
        if (!contiguous) {
            std::vector<uint32_t> part(n_regions);
            assign_lpt(cells, n_handles, part.data());
            lists.resize(n_handles);
            for (uint32_t g = 0; g < n_regions; ++g) lists[part[g]].push_back(g);
        }
        std::vector<int> status(n_handles, PHMM_OK);
        std::vector<std::string> errs(n_handles);
        std::vector<std::thread> workers;
        workers.reserve(n_handles);
        struct JoinAll {  // if starting a thread throws, the ones already running are joined before the exception travels on
            std::vector<std::thread> &w;
            ~JoinAll() {
                for (auto &t : w)
                    if (t.joinable()) t.join();
            }
        } join_all{workers};
        for (uint32_t k = 0; k < n_handles; ++k) {
            if (contiguous ? first[k] == first[k + 1] : lists[k].empty()) continue;
            workers.emplace_back([&, k] {
                phmm_handle *h = handles[k];
                try {
                    pin_near_device(h->device);
                    h->err_code = PHMM_OK;
                    status[k] = contiguous
                                    ? compute_range(h, first[k], first[k + 1], region_read_off, region_hap_off, read_off, read_bases,
                                                    base_q, ins_q, del_q, gcp, hap_off, hap_bases, out_off, out)
                                    : compute_list(h, lists[k].data(), (uint32_t)lists[k].size(), region_read_off, region_hap_off,
                                                   read_off, read_bases, base_q, ins_q, del_q, gcp, hap_off, hap_bases, out_off, out);
                    if (status[k] != PHMM_OK) errs[k] = h->err;
                } catch (const std::bad_alloc &) {
                    status[k] = PHMM_ERR_NO_MEMORY;
                    errs[k] = "phmm_compute_multi: out of host memory";
                } catch (const std::exception &e) {
                    status[k] = PHMM_ERR_INTERNAL;
                    errs[k] = std::string("phmm_compute_multi: ") + e.what();
                }
            });
        }
        for (auto &w : workers)
            if (w.joinable()) w.join();
        for (uint32_t k = 0; k < n_handles; ++k)
            if (status[k] != PHMM_OK) {
                h0->err = errs[k];
                return status[k];
            }
        return PHMM_OK;
    } catch (const std::bad_alloc &) {
        h0->err = "phmm_compute_multi: out of host memory";
        return PHMM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        h0->err = std::string("phmm_compute_multi: ") + e.what();
        return PHMM_ERR_INTERNAL;
    }
}

// The engine-level call (phmm_engine_compute) over several engines: contiguous cell-balanced ranges, every range's offsets
// rebased and its pointers moved on, nothing gathered; the whole call is validated by the first range's engine entry point
// range by range (phmm_engine_compute checks what it is given), the offsets once here.
int phmm_engine_compute_multi(phmm_handle *const *handles, uint32_t n_handles, const phmm_engine_config *cfg, uint32_t n_regions,
                              const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                              const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q,
                              const uint8_t *mapq, const uint32_t *hap_off, const uint8_t *hap_bases, const int32_t *region_ref_hap,
                              const uint64_t *out_off, double *out, uint8_t *keep) {
    if (!handles || !n_handles || !handles[0] || !cfg) return PHMM_ERR_INVALID_ARG;
    phmm_handle *h0 = handles[0];
    for (uint32_t k = 0; k < n_handles; ++k)
        if (!handles[k]) {
            h0->err = "phmm_engine_compute_multi: null handle";
            return PHMM_ERR_INVALID_ARG;
        }
    clear_thread_error(h0);
    if (n_handles == 1)
        return phmm_engine_compute(h0, cfg, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, mapq, hap_off,
                                   hap_bases, region_ref_hap, out_off, out, keep);
    if (const char *bad = validate_offsets(n_regions, region_read_off, region_hap_off, read_off, hap_off, out_off, nullptr)) {
        h0->err = bad;
        return PHMM_ERR_INVALID_ARG;
    }
    const uint32_t n_reads = region_read_off[n_regions];
    if (cfg->pcr_error_model > 3 || (read_off[n_reads] && (!read_bases || !base_q)) || (n_reads && (!mapq || !keep)) ||
        (hap_off[region_hap_off[n_regions]] && !hap_bases) || (out_off[n_regions] && !out)) {
        h0->err = cfg->pcr_error_model > 3 ? "phmm_engine_compute: Unknown PCR Error Model" : "phmm_engine_compute_multi: null pointer";
        return PHMM_ERR_INVALID_ARG;
    }
    try {
        std::vector<uint32_t> first(n_handles + 1);
        split_contiguous(region_cells(n_regions, region_read_off, region_hap_off, read_off, hap_off), n_handles, first.data());
        std::vector<int> st(n_handles, PHMM_OK);
        std::vector<std::string> errs(n_handles);
        std::vector<std::thread> workers;
        workers.reserve(n_handles);
        struct JoinAll {
            std::vector<std::thread> &w;
            ~JoinAll() {
                for (auto &t : w)
                    if (t.joinable()) t.join();
            }
        } join_all{workers};
        for (uint32_t k = 0; k < n_handles; ++k) {
            if (first[k] == first[k + 1]) continue;
            workers.emplace_back([&, k] {
                phmm_handle *h = handles[k];
                try {
                    pin_near_device(h->device);
                    ChunkView c;
                    c.g1 = first[k];
                    (void)next_chunk(c, first[k + 1], region_read_off, region_hap_off, read_off, hap_off, out_off, true);
                    const size_t bo = c.read_byte0, co = c.hap_byte0;
                    st[k] = phmm_engine_compute(h, cfg, c.g1 - c.g0, c.rro.data(), c.rho.data(), c.ro.data(), read_bases + bo, base_q + bo,
                                                ins_q ? ins_q + bo : nullptr, del_q ? del_q + bo : nullptr, mapq + c.r0, c.ho.data(),
                                                hap_bases + co, region_ref_hap ? region_ref_hap + c.g0 : nullptr, c.oo.data(),
                                                out + out_off[c.g0], keep + c.r0);
                    if (st[k] != PHMM_OK) errs[k] = h->err;
                } catch (const std::bad_alloc &) {
                    st[k] = PHMM_ERR_NO_MEMORY;
                    errs[k] = "phmm_engine_compute_multi: out of host memory";
                } catch (const std::exception &e) {
                    st[k] = PHMM_ERR_INTERNAL;
                    errs[k] = std::string("phmm_engine_compute_multi: ") + e.what();
                }
            });
        }
        for (auto &w : workers)
            if (w.joinable()) w.join();
        for (uint32_t k = 0; k < n_handles; ++k)
            if (st[k] != PHMM_OK) {
                h0->err = errs[k];
                return st[k];
            }
        return PHMM_OK;
    } catch (const std::bad_alloc &) {
        h0->err = "phmm_engine_compute_multi: out of host memory";
        return PHMM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        h0->err = std::string("phmm_engine_compute_multi: ") + e.what();
        return PHMM_ERR_INTERNAL;
    }
}

// The whole per-region path (phmm_region_compute) over several engines: contiguous cell-balanced ranges of regions, one
// host thread per engine pinned next to its GPU, every range staged straight from the caller's arrays and its results
// written where phmm_region_compute on one engine would write them.
int phmm_region_compute_multi(phmm_handle *const *handles, uint32_t n_handles, const phmm_engine_config *cfg, const phmm_realign_config *rcfg,
                              uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off, const uint32_t *read_off,
                              const uint8_t *read_bases, const uint8_t *base_q, const uint8_t *ins_q, const uint8_t *del_q, const uint8_t *mapq,
                              const uint32_t *read_soft_clip, const uint32_t *hap_off, const uint8_t *hap_bases, const int32_t *region_ref_hap,
                              const uint64_t *out_off, const int32_t *hap_priority, const uint64_t *region_reference_start,
                              const uint32_t *hap_cigar_off, const uint32_t *hap_cigar, const uint32_t *hap_start_wrt_ref,
                              const uint32_t *orig_cigar_off, const uint32_t *orig_cigar, const uint64_t *out_cigar_off, double *out, uint8_t *keep,
                              int32_t *best_allele, double *likelihood, double *confidence, uint32_t *out_cigar, uint32_t *n_out_cigar,
                              int64_t *new_pos, int32_t *status) {
    if (!handles || !n_handles || !handles[0] || !cfg || !rcfg) return PHMM_ERR_INVALID_ARG;
    phmm_handle *h0 = handles[0];
    for (uint32_t k = 0; k < n_handles; ++k)
        if (!handles[k]) {
            h0->err = "phmm_region_compute_multi: null handle";
            return PHMM_ERR_INVALID_ARG;
        }
    clear_thread_error(h0);
    try {
        const RegionArgs a = region_pack_args(cfg, rcfg, n_regions, region_read_off, region_hap_off, read_off, read_bases, base_q, ins_q, del_q, mapq,
                                              read_soft_clip, hap_off, hap_bases, region_ref_hap, out_off, hap_priority, region_reference_start,
                                              hap_cigar_off, hap_cigar, hap_start_wrt_ref, orig_cigar_off, orig_cigar, out_cigar_off, out, keep,
                                              best_allele, likelihood, confidence, out_cigar, n_out_cigar, new_pos, status);
        const std::string bad = region_validate(a);  // the whole call, before any engine indexes anything
        if (!bad.empty()) {
            h0->err = bad;
            return h0->err_code = PHMM_ERR_INVALID_ARG;
        }
        if (n_handles == 1) {
            h0->err_code = PHMM_OK;
            return region_compute(h0, a);
        }
        std::vector<uint32_t> first(n_handles + 1);
        split_contiguous(region_cells(n_regions, region_read_off, region_hap_off, read_off, hap_off), n_handles, first.data());
        std::vector<int> st(n_handles, PHMM_OK);
        std::vector<std::string> errs(n_handles);
        std::vector<std::thread> workers;
        workers.reserve(n_handles);
        struct JoinAll {
            std::vector<std::thread> &w;
            ~JoinAll() {
                for (auto &t : w)
                    if (t.joinable()) t.join();
            }
        } join_all{workers};
        for (uint32_t k = 0; k < n_handles; ++k) {
            if (first[k] == first[k + 1]) continue;
            workers.emplace_back([&, k] {
                phmm_handle *h = handles[k];
                try {
                    pin_near_device(h->device);
                    h->err_code = PHMM_OK;
                    st[k] = region_compute_range(h, a, first[k], first[k + 1]);
                    if (st[k] != PHMM_OK) errs[k] = h->err;
                } catch (const std::bad_alloc &) {
                    st[k] = PHMM_ERR_NO_MEMORY;
                    errs[k] = "phmm_region_compute_multi: out of host memory";
                } catch (const std::exception &e) {
                    st[k] = PHMM_ERR_INTERNAL;
                    errs[k] = std::string("phmm_region_compute_multi: ") + e.what();
                }
            });
        }
        for (auto &w : workers)
            if (w.joinable()) w.join();
        // the first failure in region order; a CIGAR slot that is too small on one engine does not hide a real failure on another
        int worst = PHMM_OK;
        for (uint32_t k = 0; k < n_handles; ++k)
            if (st[k] != PHMM_OK && (worst == PHMM_OK || worst == PHMM_ERR_CIGAR_CAPACITY)) {
                worst = st[k];
                h0->err = errs[k];
            }
        return h0->err_code = worst;
    } catch (const std::bad_alloc &) {
        h0->err = "phmm_region_compute_multi: out of host memory";
        return PHMM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        h0->err = std::string("phmm_region_compute_multi: ") + e.what();
        return PHMM_ERR_INTERNAL;
    }
}

}  // extern "C"
