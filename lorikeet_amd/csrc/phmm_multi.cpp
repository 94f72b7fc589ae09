// phmm_assign_regions / phmm_compute_multi (include/phmm.h): one call over several engines, one per device, from one
// process -- what a single Lorikeet process on a multi-GPU node needs.  Whole regions are assigned by greedy
// longest-processing-time on cells(region); every engine computes its share concurrently, on a host thread of its own,
// into disjoint slices of `out`.  There is no exchange between devices and no CPU fallback.
#include <algorithm>
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

struct Share {  // the regions of one engine, gathered into arrays of their own (ascending region order)
    std::vector<uint32_t> regions, rro{0}, rho{0}, ro{0}, ho{0};
    std::vector<uint64_t> oo{0};
    std::vector<uint8_t> bytes[6];
    std::vector<double> out;
    int status = PHMM_OK;
    std::string err;
};

}  // namespace

extern "C" {

int phmm_assign_regions(uint32_t n_regions, const uint32_t *region_read_off, const uint32_t *region_hap_off,
                        const uint32_t *read_off, const uint32_t *hap_off, uint32_t n_parts, uint32_t *part_of_region) {
    if (!n_parts || (n_regions && (!region_read_off || !region_hap_off || !read_off || !hap_off || !part_of_region)))
        return PHMM_ERR_INVALID_ARG;
    assign_lpt(region_cells(n_regions, region_read_off, region_hap_off, read_off, hap_off), n_parts, part_of_region);
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
    std::vector<uint32_t> part(n_regions);
    assign_lpt(region_cells(n_regions, region_read_off, region_hap_off, read_off, hap_off), n_handles, part.data());
    std::vector<Share> shares(n_handles);
    const uint8_t *src[6] = {read_bases, base_q, ins_q, del_q, gcp, hap_bases};
    for (uint32_t g = 0; g < n_regions; ++g) {
        Share &s = shares[part[g]];
        s.regions.push_back(g);
        const uint32_t r0 = region_read_off[g], r1 = region_read_off[g + 1], a0 = region_hap_off[g], a1 = region_hap_off[g + 1];
        for (uint32_t r = r0; r < r1; ++r) s.ro.push_back(s.ro.back() + (read_off[r + 1] - read_off[r]));
        for (uint32_t a = a0; a < a1; ++a) s.ho.push_back(s.ho.back() + (hap_off[a + 1] - hap_off[a]));
        for (int i = 0; i < 5; ++i) s.bytes[i].insert(s.bytes[i].end(), src[i] + read_off[r0], src[i] + read_off[r1]);
        s.bytes[5].insert(s.bytes[5].end(), hap_bases + hap_off[a0], hap_bases + hap_off[a1]);
        s.rro.push_back(s.rro.back() + (r1 - r0));
        s.rho.push_back(s.rho.back() + (a1 - a0));
        s.oo.push_back(s.oo.back() + (out_off[g + 1] - out_off[g]));
    }
    std::vector<std::thread> workers;
    for (uint32_t k = 0; k < n_handles; ++k) {
        Share &s = shares[k];
        if (s.regions.empty()) continue;
        s.out.resize(s.oo.back());
        workers.emplace_back([&s, h = handles[k]] {
            s.status = phmm_compute(h, (uint32_t)s.regions.size(), s.rro.data(), s.rho.data(), s.ro.data(), s.bytes[0].data(),
                                    s.bytes[1].data(), s.bytes[2].data(), s.bytes[3].data(), s.bytes[4].data(), s.ho.data(),
                                    s.bytes[5].data(), s.oo.data(), s.out.data());
            if (s.status != PHMM_OK) s.err = phmm_last_error(h);
        });
    }
    for (auto &w : workers) w.join();
    int st = PHMM_OK;
    for (uint32_t k = 0; k < n_handles; ++k) {
        Share &s = shares[k];
        // a share's results are valid numbers even when one of them tripped the `<= 0` check: hand everything over
        if (s.status == PHMM_OK || s.status == PHMM_ERR_POSITIVE_RESULT)
            for (size_t i = 0; i < s.regions.size(); ++i) {
                const uint32_t g = s.regions[i];
                const uint64_t n = s.oo[i + 1] - s.oo[i];
                if (n) memcpy(out + out_off[g], s.out.data() + s.oo[i], n * 8);
            }
        if (s.status != PHMM_OK && st == PHMM_OK) {
            st = s.status;
            h0->err = s.err;
        }
    }
    return st;
}

}  // extern "C"
