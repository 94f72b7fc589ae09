"""Developer tool: random regions (reference + variant haplotypes with exact CIGARs, reads with errors and clips) through
phmm_compute -> phmm_realign_to_best -> phmm_project_to_reference, every read against oracle/cigar_oracle.c (status,
position, CIGAR equal).  usage: python tools/soak_project.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lorikeet_amd import HipPairHMMEngine, realign  # noqa: E402
from lorikeet_amd.smith_waterman import ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, SmithWatermanAligner  # noqa: E402
from oracle import oracle  # noqa: E402
from project_scenarios import oracle_read, scenario  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = HipPairHMMEngine(0)
t0 = time.time()
n_batches = n_reads = 0
statuses = {}
while time.time() - t0 < seconds:
    b, hap_cigars, hap_starts, ref_hap, ref_start, orig = scenario(seed * 100003 + n_batches, n_regions=8, low_complexity=n_batches % 3 == 0)
    reg = np.repeat(np.arange(b.n_regions), np.diff(b.region_read_off.astype(np.int64)))
    if n_batches % 4 == 3:  # any haplotype of the region instead of the best one, some reads without one: the hopeless pairs too
        rng = np.random.default_rng(n_batches)
        which = rng.integers(-1, np.diff(b.region_hap_off.astype(np.int64))[reg]).astype(np.int32)
        idx = np.where(which >= 0, b.region_hap_off[:-1].astype(np.int64)[reg] + which, -1)
        haps = [b.hap_bases[int(b.hap_off[a]):int(b.hap_off[a + 1])] for a in range(b.n_haps)]
        reads = [b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])] for r in range(b.n_reads)]
        aligned = SmithWatermanAligner(eng).align_indexed(haps, reads, idx, ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS, "SoftClip")
    else:
        best, aligned = realign.realign_reads_to_their_best_haplotype(eng, b, eng.compute(b))
        which = best.allele_index
    got = realign.project_to_reference(eng, b, which, aligned, hap_cigars, hap_starts, ref_hap, ref_start, orig)
    for r in range(b.n_reads):
        st, pos, cig = oracle_read(b, r, reg[r], which[r], hap_cigars, hap_starts, ref_hap, ref_start, orig)
        assert got.status[r] == st, (n_batches, r, got.status[r], st)
        if st == 0:
            assert got.new_pos[r] == pos and oracle.cigar_to_string(got.cigars[r]) == cig, (n_batches, r, oracle.cigar_to_string(got.cigars[r]), cig)
        statuses[st] = statuses.get(st, 0) + 1
    n_batches += 1
    n_reads += b.n_reads
print("projection soak ok: %d batches, %d reads, status, position and CIGAR equal to the oracle; statuses %s" % (n_batches, n_reads, statuses))
