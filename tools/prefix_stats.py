"""Prefix statistics of the bench workloads: how much of a region's PairHMM work lies in haplotype
prefixes shared with another haplotype of the region -- what the reference's scalar arm skips through
find_first_position_where_haplotypes_differ (pair_hmm.rs:452-464,706-717) -- and what a trunk + suffix scheme on the
16-lanes-x-K mapping could save: a suffix that starts at a lane boundary of its trunk (a multiple of the trunk's K columns)
runs as a pair of its own with K' = ceil(suffix / 16) columns per lane, cost (7 K' + 11) / (7 K + 11) of a full pair.
usage: python tools/prefix_stats.py [config2 config3 config5 ragged]"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from lorikeet_amd import synthetic  # noqa: E402


def lcp(a, b):
    n = min(len(a), len(b))
    d = np.flatnonzero(a[:n] != b[:n])
    return int(d[0]) if len(d) else n


def region_stats(haps, sum_r):
    """haps: list of byte arrays.  Returns cells (full), cells the reference's next-haplotype rule skips (consecutive order),
    cells a prefix tree skips (every column shared with an EARLIER haplotype), and the modelled kernel cost ratio."""
    H = [len(h) for h in haps]
    full = sum_r * sum(H)
    consecutive = sum(lcp(haps[k - 1], haps[k]) for k in range(1, len(haps))) * sum_r
    best = [max(lcp(haps[j], haps[k]) for j in range(k)) for k in range(1, len(haps))]
    tree = sum(best) * sum_r
    cost_full = cost_split = 0.0
    for k, h in enumerate(haps):
        K = max(2, -(-len(h) // 16))
        c = 7 * K + 11
        cost_full += c
        if k == 0 or len(h) != len(haps[0]):   # (the scheme needs equal D(0, j) = 2^1020 / H: equal lengths, or a rescale)
            cost_split += c
            continue
        s = best[k - 1] // K                    # whole lanes of the trunk in front of the first difference
        rest = len(h) - s * K
        K2 = max(2, -(-rest // 16))
        cost_split += min(c, 7 * K2 + 11 + 1.5)  # (+ the head lane's boundary select)
    return full, consecutive, tree, cost_full, cost_split


STREAM_EFFICIENCY = {1: 1.0, 2: 0.93, 4: 0.83}  # measured: 3990 / 3700 / 3300 GCUPS at 1 / 2 / 4 streams of reads per wave (phmm_api.cpp)


def scheme_cost(haps, S):
    """What the scheme that fits the kernel costs, in (7 K + 11)-instruction steps per read row of a run: haplotypes sorted by
    the length of the prefix they share with the region's first one, 4 / S of them per wave (S streams of reads side by side),
    every wave's suffix starting at the lane boundary its LEAST sharing member allows; the first haplotype and the remainder
    stay whole, and the trunk pays ~6 instructions per step for parking its column."""
    nh = len(haps)
    K = max(2, -(-max(len(h) for h in haps) // 16))
    full = -(-nh // 4) * (7 * K + 11)
    if nh < 2:
        return full, full
    P = [lcp(haps[0], haps[k]) for k in range(1, nh)]
    order = sorted(range(1, nh), key=lambda k: -P[k - 1])
    gs = 4 // S
    groups = [order[j:j + gs] for j in range(0, len(order), gs)]
    tot, root_placed = 0.0, False
    for g in groups:
        if len(g) < gs and not root_placed:
            tot += (7 * K + 17) / S / STREAM_EFFICIENCY[S]
            root_placed = True
            continue
        s = min(P[k - 1] for k in g) // K
        K2 = max(2, -(-(max(len(haps[k]) for k in g) - s * K) // 16))
        c = 7 * K2 + 17 if s >= 1 and 7 * K2 + 17 < 7 * K + 11 else 7 * K + 11
        tot += c / S / STREAM_EFFICIENCY[S]
    if not root_placed:
        tot += (7 * K + 17) / S / STREAM_EFFICIENCY[S]
    return full, min(tot, full)


def main():
    for name in sys.argv[1:] or ["config2", "config3", "config5", "ragged"]:
        if name == "ragged":
            b = synthetic.ragged(1536)
        elif name == "config2":
            b = synthetic.make_regions(256, 128, 8, 300, [150], seed=1000)
        else:
            n = {"config3": 512, "config5": 32}[name]
            b = synthetic.config(name, only=(0, n))
        tot = np.zeros(5)
        sch = np.zeros((3, 2))
        for g in range(b.n_regions):
            h0, h1 = int(b.region_hap_off[g]), int(b.region_hap_off[g + 1])
            haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(h0, h1)]
            r0, r1 = int(b.region_read_off[g]), int(b.region_read_off[g + 1])
            sum_r = int(b.read_off[r1]) - int(b.read_off[r0])
            tot += np.array(region_stats(haps, sum_r)) * np.array([1, 1, 1, sum_r, sum_r])
            for i, S in enumerate((1, 2, 4)):
                sch[i] += np.array(scheme_cost(haps, S)) * sum_r
        full, cons, tree, cf, cs = tot
        print("%-8s %5d regions: cells shared with the previous haplotype %.3f, with any earlier one %.3f; "
              "ideal trunk + suffix cost (every suffix a pair of its own) %.3f of full -> effective x %.3f; the scheme that fits the "
              "kernel, 1 / 2 / 4 streams per wave: x %.3f / %.3f / %.3f" % (name, b.n_regions, cons / full, tree / full, cs / cf, cf / cs,
                                                                           *(sch[i][0] / sch[i][1] for i in range(3))))


if __name__ == "__main__":
    main()
