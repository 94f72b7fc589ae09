"""Developer tool: what the shapes of a bench workload cost the PairHMM kernels in padding, from the LIBRARY'S
OWN PLAN of the batch -- phmm_plan_describe, host only, no GPU: swept lane-cells = steps x 64 lanes x K columns of every wave
(= phmm_batch_executed_cells, rows.<workload>.executed_per_cell of the bench line), split into
  columns     lanes x K - H per pair: the columns beyond a haplotype's end
  slots       haplotype slots a wave leaves empty (a region's haplotype count against 4 / 2 / 1 slots per wave or stream)
  steps       steps that carry no read row: the SUM / RESET rows between the reads of a run, the L - 1 steps a run needs to reach its
              last lane, what the longest stream of a wave has more than the others.
The whole batch first (the plan the bench runs), then the regions by K = ceil(longest haplotype / 16) -- each range planned as a
batch of its own, so the ranges' plans need not add up to the whole batch's exactly.
usage: python tools/ragged_padding.py [ragged|config2|config3|config5]  ->  the table of NOTEBOOK.md 20.6"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lorikeet_amd import synthetic  # noqa: E402
from lorikeet_amd.engine import plan_describe  # noqa: E402

RANGES = [(2, 9), (10, 15), (16, 19), (20, 25), (26, 99)]


def subset(b, regions):
    """The regions `regions` of b as a batch of their own (offsets only: the planner reads nothing else)."""
    rro, rho, ro, ho = (x.astype(np.int64) for x in (b.region_read_off, b.region_hap_off, b.read_off, b.hap_off))
    rl, hl, nr, nh = [], [], [0], [0]
    for g in regions:
        rl.append(np.diff(ro[rro[g]:rro[g + 1] + 1]))
        hl.append(np.diff(ho[rho[g]:rho[g + 1] + 1]))
        nr.append(nr[-1] + len(rl[-1]))
        nh.append(nh[-1] + len(hl[-1]))
    class Offsets:  # (what plan_describe reads of a RegionBatch)
        pass
    s = Offsets()
    s.region_read_off = np.asarray(nr, np.uint32)
    s.region_hap_off = np.asarray(nh, np.uint32)
    s.read_off = np.concatenate([[0], np.cumsum(np.concatenate(rl))]).astype(np.uint32)
    s.hap_off = np.concatenate([[0], np.cumsum(np.concatenate(hl))]).astype(np.uint32)
    s.n_regions = len(regions)
    return s


def line(name, n, info):
    u = info.cells
    steps = info.swept_cells - u - info.pad_column_cells - info.pad_slot_cells
    print("%-10s %8d %14.3e %9.1f%% %9.1f%% %9.1f%% %12.3f   %s" % (name, n, u, 100 * info.pad_column_cells / u, 100 * info.pad_slot_cells / u,
                                                                 100 * steps / u, info.swept_cells / u, info.dominant_kernel.decode()))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "ragged"
    b = {"ragged": synthetic.ragged, "config2": lambda: synthetic.config2(1024), "config3": synthetic.config3, "config5": synthetic.config5}[what]()
    ho, rho = b.hap_off.astype(np.int64), b.region_hap_off.astype(np.int64)
    print("%s: %d regions; padding in %% of the useful cells, from phmm_plan_describe" % (what, b.n_regions))
    print("%-10s %8s %14s %10s %10s %10s %12s   %s" % ("K range", "regions", "useful cells", "columns", "slots", "steps", "swept/useful", "dominant kernel"))
    line("all", b.n_regions, plan_describe(b))
    by_range = {r: [] for r in RANGES}
    for g in range(b.n_regions):
        H = np.diff(ho[rho[g]:rho[g + 1] + 1])
        if len(H) and b.region_read_off[g + 1] > b.region_read_off[g]:
            K = max(2, int(-(-H.max() // 16)))
            by_range[next(r for r in RANGES if r[0] <= K <= r[1])].append(g)
    for r, regions in by_range.items():
        if regions:
            line("%d-%d" % r, len(regions), plan_describe(subset(b, regions)))


if __name__ == "__main__":
    main()
