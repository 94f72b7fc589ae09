"""Developer tool: what the shapes of the bench's ragged mix cost the chained PairHMM kernels in padding, by range of K
(VERDICT r5 item 3).  Per region the planner sweeps 16 lanes x K = ceil(H / 16) columns per haplotype and four haplotype slots per
wave: columns beyond a haplotype's end and slots a wave leaves empty are swept like real cells.  Counted here from the shapes alone
(no GPU): useful cells = sum R x H over the pairs; column padding = R x (16 K - H); slot padding = the empty slots of a region's last
wave where the remainder is not re-packed into multi-stream items (1 or 2 of 4 slots filled twice or four times over); per-read
rows = the SUM / RESET rows and the 15 fill steps a run of reads pays once (CHAIN_MAX_READS = 64 reads per run at most).
usage: python tools/ragged_padding.py  ->  the table of NOTEBOOK.md 20.6"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lorikeet_amd import synthetic  # noqa: E402

RANGES = [(2, 9), (10, 15), (16, 19), (20, 25), (26, 99)]


def main():
    b = synthetic.ragged()
    rro, rho, ro, ho = (x.astype(np.int64) for x in (b.region_read_off, b.region_hap_off, b.read_off, b.hap_off))
    rows = {r: dict(regions=0, useful=0, cols=0, slots=0, steps=0, swept=0) for r in RANGES}
    for g in range(b.n_regions):
        R = np.diff(ro[rro[g]:rro[g + 1] + 1])
        H = np.diff(ho[rho[g]:rho[g + 1] + 1])
        if not len(R) or not len(H):
            continue
        L = 16 if H.max() <= 400 else 32 if H.max() <= 800 else 64
        K = max(2, int(-(-H.max() // L)))
        per_wave = 64 // L
        rng = next(r for r in RANGES if r[0] <= K <= r[1])
        sum_r, nh = int(R.sum()), len(H)
        useful = sum_r * int(H.sum())
        cols = sum_r * int((L * K - H).sum())                      # columns beyond each haplotype's end
        rem = nh % per_wave
        # (a remainder of 1 or 2 haplotypes at 16 lanes -- a whole region of 1 or 2 as well -- is swept as 4 or 2 streams of reads
        # side by side: no empty slot; a remainder of 3 leaves one slot of its wave empty)
        empty = 0 if rem == 0 or (L == 16 and rem in (1, 2)) else per_wave - rem
        slots = sum_r * empty * L * K
        waves = -(-nh // per_wave)
        runs = -(-len(R) // 64)
        steps = waves * (2 * len(R) + (L - 1) * runs) * 64 * K    # SUM / RESET rows per read, fill steps per run of reads
        d = rows[rng]
        d["regions"] += 1
        d["useful"] += useful
        d["cols"] += cols
        d["slots"] += slots
        d["steps"] += steps
        d["swept"] += useful + cols + slots + steps
    tot = {k: sum(d[k] for d in rows.values()) for k in ("regions", "useful", "cols", "slots", "steps", "swept")}
    print("ragged mix: %d regions, %.3e useful cells" % (b.n_regions, tot["useful"]))
    print("%-10s %8s %14s %10s %10s %12s %12s" % ("K range", "regions", "useful cells", "columns", "slots", "extra rows", "swept/useful"))
    for r, d in list(rows.items()) + [("all", tot)]:
        if not d["useful"]:
            continue
        u = d["useful"]
        print("%-10s %8d %14.3e %9.1f%% %9.1f%% %11.1f%% %12.3f" % ("%d-%d" % r if r != "all" else "all", d["regions"], u, 100 * d["cols"] / u,
                                                                100 * d["slots"] / u, 100 * d["steps"] / u, d["swept"] / u))


if __name__ == "__main__":
    main()
