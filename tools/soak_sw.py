"""Developer tool: Smith-Waterman soak -- random batches of pairs of widely varying shape, all four overhang strategies,
random parameter sets, through phmm_sw_align vs the oracle (the reference's scalar arm in C): CIGAR and offset must be equal.
usage: python tools/soak_sw.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lorikeet_amd import HipPairHMMEngine  # noqa: E402
from lorikeet_amd.smith_waterman import Parameters, SmithWatermanAligner  # noqa: E402
from oracle import oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
eng = HipPairHMMEngine(0)
al = SmithWatermanAligner(eng)
alpha = b"ACGT"


def rnd(n, k=4):
    return bytes(alpha[int(x)] for x in rng.integers(0, k, n))


def mutate(seq):
    out, i = [], 0
    ps, pi = rng.choice([0.0, 0.01, 0.05, 0.2]), rng.choice([0.0, 0.005, 0.03])
    while i < len(seq):
        u = rng.random()
        if u < pi / 2:
            i += int(rng.integers(1, 12))
        elif u < pi:
            out.extend(rnd(int(rng.integers(1, 12))))
        elif u < pi + ps:
            out.append(alpha[int(rng.integers(0, 4))])
            i += 1
        else:
            out.append(seq[i])
            i += 1
    return bytes(out) or b"A"


t_end = time.time() + budget
n_batches = n_pairs = n_cells = 0
kinds = {}
modes = {}
while time.time() < t_end:
    kind = str(rng.choice(["reads", "reads", "haps", "tiny", "unrelated", "long", "lowcomplexity"]))
    pairs = []
    for _ in range(int(rng.integers(1, 300)) if kind != "long" else int(rng.integers(1, 6))):
        if kind == "reads":
            ref = rnd(int(rng.integers(60, 520)))
            s = int(rng.integers(0, len(ref)))
            alt = mutate(ref[s:s + int(rng.integers(20, 260))])
        elif kind == "haps":
            ref = rnd(int(rng.integers(100, 700)))
            alt = mutate(ref)
        elif kind == "tiny":
            ref, alt = rnd(int(rng.integers(1, 20))), rnd(int(rng.integers(1, 20)))
        elif kind == "unrelated":
            ref, alt = rnd(int(rng.integers(1, 400))), rnd(int(rng.integers(1, 400)))
        elif kind == "long":
            ref = rnd(int(rng.integers(800, 3000)))
            alt = mutate(ref[int(rng.integers(0, 300)):])
        else:
            ref, alt = rnd(int(rng.integers(5, 300)), 2), rnd(int(rng.integers(5, 300)), 2)
        pairs.append((ref, alt))
    ext = int(rng.integers(1, 12))
    prm = [Parameters(3, -1, -4, -3), Parameters(25, -50, -110, -6), Parameters(200, -150, -260, -11), Parameters(10, -15, -30, -5),
           Parameters(int(rng.integers(1, 30)), -int(rng.integers(1, 30)), -(ext + int(rng.integers(0, 40))), -ext)][int(rng.integers(0, 5))]
    strategy = str(rng.choice(["SoftClip", "InDel", "LeadingInDel", "Ignore"]))
    # how the call is cut and whether it takes the tags-only first pass: the planner's choice, or forced either way
    lite, chunks = int(rng.choice([-1, -1, 0, 1, 1])), int(rng.choice([0, 0, 1, 2, 3]))
    eng.set_switch("sw_lite", lite)
    eng.set_switch("sw_chunks", chunks)
    modes[(lite, chunks)] = modes.get((lite, chunks), 0) + 1
    got = al.align_batch(pairs, prm, strategy, capacity=int(rng.choice([4, 24, 200])))
    p4 = [prm.match_value, prm.mismatch_penalty, prm.gap_open_penalty, prm.gap_extend_penalty]
    for g, (ref, alt) in zip(got, pairs):
        cig, off = oracle.sw_align(ref, alt, p4, strategy)
        assert g.alignment_offset == off and np.array_equal(g.elements, cig), (kind, strategy, p4, ref, alt, g, oracle.cigar_to_string(cig), off)
        n_cells += len(ref) * len(alt)
    n_batches += 1
    n_pairs += len(pairs)
    kinds[kind] = kinds.get(kind, 0) + 1
print("sw soak ok: %d batches, %d alignments, %.3g cells, every CIGAR and offset equal to the oracle; kinds %s; "
      "(sw_lite, sw_chunks) switches drawn per batch: %d combinations" % (n_batches, n_pairs, n_cells, kinds, len(modes)))
