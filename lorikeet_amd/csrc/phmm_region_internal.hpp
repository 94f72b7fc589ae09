// Private to the per-region pipeline (phmm_region.cpp: kernels launched per call; phmm_server.cpp: the same steps as tasks of
// the resident region server): where the pieces of one call lie in an arena.
#pragma once
#include <cstddef>
#include <cstring>

#include "phmm_host.hpp"

namespace phmm_host {

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// Where everything of one enqueue lies in the slot's arena (and, at the same offsets, in its pinned mirror), behind the
// batch's own metadata: [inputs ... status_in] travel to the device, [q ... swo] exist on the device only, [res ... end)
// come back.  Every piece starts on a 256-byte boundary.
struct Layout {
    size_t bases, q0, i0, d0, mapq, haps, refhap, pri, rstart, hco, hc, hs, oco, oc, outco, clip, status_in, in_end;
    size_t q, i, d, g, thr, refidx, swc, nsw, swo, todo;
    size_t res, keep, out, best, lk, conf, pst, pno, pos, pout, end;
    Layout() { memset(this, 0, sizeof *this); }
    // pair_stride > 0: the aligner's slots are one per (read, haplotype of its region), pair_stride of them per read
    Layout(size_t base, const RegionArgs &a, uint32_t sw_capacity, uint32_t pair_stride) {
        const uint32_t ng = a.n_regions, nr = a.region_read_off[ng], nh = a.region_hap_off[ng];
        const size_t n_sw = pair_stride ? (size_t)nr * pair_stride : nr;
        const size_t rb = a.read_off[nr], hb = a.hap_off[nh];
        size_t used = base;
        auto take = [&](size_t bytes) {
            const size_t off = up256(used);
            used = off + bytes;
            return off;
        };
        bases = take(rb);
        q0 = take(rb);
        i0 = take(a.ins_q ? rb : 0);
        d0 = take(a.del_q ? rb : 0);
        mapq = take(nr);
        haps = take(hb);
        refhap = take(4ull * ng);
        pri = take(a.hap_priority ? 4ull * nh : 0);
        rstart = take(8ull * ng);
        hco = take(4ull * (nh + 1));
        hc = take(4ull * a.hap_cigar_off[nh]);
        hs = take(4ull * nh);
        oco = take(4ull * (nr + 1));
        oc = take(4ull * a.orig_cigar_off[nr]);
        outco = take(8ull * (nr + 1));
        clip = take(a.read_soft_clip ? 8ull * nr : 0);
        status_in = take(256);
        in_end = up256(used);
        q = take(rb);
        i = take(rb);
        d = take(rb);
        g = take(rb);
        thr = take(8ull * nr);
        refidx = take(4ull * nr);
        swc = take(4ull * n_sw * sw_capacity);
        nsw = take(4ull * n_sw);
        swo = take(4ull * n_sw);
        todo = take(4ull * nr);  // the list the aligner's tags-only pass leaves to its second pass (SW_LITE)
        res = take(256);
        keep = take(nr);
        out = take(8ull * a.out_off[ng]);
        best = take(4ull * nr);
        lk = take(8ull * nr);
        conf = take(8ull * nr);
        pst = take(4ull * nr);
        pno = take(4ull * nr);
        pos = take(8ull * nr);
        pout = take(4ull * a.out_cigar_off[nr]);
        end = up256(used);
    }
};

}  // namespace phmm_host
