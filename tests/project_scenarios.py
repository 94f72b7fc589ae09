"""Random scenarios for the projection of reads onto the reference (tests/test_project_hip.py, tools/soak_project.py):
regions of a reference plus variant haplotypes with exact haplotype -> reference CIGARs, reads cut out of the haplotypes
with errors, original CIGARs with clips; and the oracle's answer for one read."""
import numpy as np

from lorikeet_amd.batch import Read, RegionBatch
from oracle import oracle
from oracle.oracle import CigarError

BEST = [10, -15, -30, -5]  # ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS
ALPHA = b"ACGT"


def make_read(bases):
    n = len(bases)
    return Read(bases, np.full(n, 30, np.uint8), np.full(n, 45, np.uint8), np.full(n, 45, np.uint8), np.full(n, 10, np.uint8))


def rnd(rng, n, k=4):
    return bytes(ALPHA[int(x)] for x in rng.integers(0, k, n))


def variant_haplotype(rng, reference, low_complexity):
    """A haplotype = the reference with a few SNVs / insertions / deletions; its CIGAR against the reference is built
    alongside (so it is exact), sometimes with the deletion-next-to-insertion shapes the builder has rules for."""
    hap, cigar, i = bytearray(), [], 0
    while i < len(reference):
        u = rng.random()
        step = int(rng.integers(8, 40))
        if u < 0.12 and i > 5:       # deletion
            d = int(rng.integers(1, 7))
            cigar.append((d, "D"))
            i += d
        elif u < 0.24 and i > 5:     # insertion (a repeat of what precedes it when low_complexity: left-alignable)
            k = int(rng.integers(1, 7))
            ins = bytes(hap[-k:]) if low_complexity and len(hap) >= k else rnd(rng, k)
            hap += ins
            cigar.append((k, "I"))
        seg = bytearray(reference[i:i + step])
        if len(seg) and rng.random() < 0.3:
            seg[int(rng.integers(0, len(seg)))] = ALPHA[int(rng.integers(0, 4))]
        hap += seg
        if len(seg):
            cigar.append((len(seg), "M"))
        i += step
    text = "".join("%d%s" % c for c in cigar)
    return bytes(hap), oracle.cigar_builder([text], remove_deletions_at_ends=False)[0] if text else "1M"


def scenario(seed, n_regions=6, low_complexity=False):
    rng = np.random.default_rng(seed)
    regions, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars, reads_clipped = [], [], [], [], [], [], []
    for g in range(n_regions):
        k = 2 if low_complexity else 4
        reference = rnd(rng, int(rng.integers(150, 320)), k)
        haps, cigs = [reference], ["%dM" % len(reference)]
        for _ in range(int(rng.integers(1, 5))):
            h, c = variant_haplotype(rng, reference, low_complexity)
            if h not in haps:
                haps.append(h)
                cigs.append(c)
        starts = [0] + [int(rng.integers(0, 3)) for _ in haps[1:]]  # alignment_start_hap_wrt_ref
        reads = []
        for _ in range(int(rng.integers(3, 14))):
            h = haps[int(rng.integers(0, len(haps)))]
            s = int(rng.integers(0, max(1, len(h) - 40)))
            core = bytearray(h[s:s + int(rng.integers(25, 110))])
            for _m in range(int(rng.integers(0, 3))):
                q = int(rng.integers(0, len(core)))
                u = rng.random()
                if u < 0.5:
                    core[q] = ALPHA[int(rng.integers(0, 4))]
                elif u < 0.75:
                    core[q:q] = rnd(rng, int(rng.integers(1, 5)))
                else:
                    del core[q:q + int(rng.integers(1, 5))]
            core = bytes(core) or b"A"
            lh, ls, ts, th = (int(rng.integers(0, 6)) * int(rng.random() < 0.3) for _ in range(4))
            orig = ("%dH" % lh if lh else "") + ("%dS" % ls if ls else "") + "%dM" % len(core) + ("%dS" % ts if ts else "") + ("%dH" % th if th else "")
            reads.append(make_read(core))
            orig_cigars.append(oracle.parse_cigar(orig))
            reads_clipped.append(core)
        regions.append((reads, haps))
        hap_cigars += [oracle.parse_cigar(c) for c in cigs]
        hap_starts += starts
        ref_hap.append(0)
        ref_start.append(int(rng.integers(1, 10 ** 9)))
    return RegionBatch.from_regions(regions), hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars


def oracle_read(b, r, g, best, hap_cigars, hap_starts, ref_hap, ref_start, orig_cigars):
    """(status, pos, cigar string) the oracle gives read r when it is aligned to haplotype `best` of its region."""
    if best < 0:
        return 1, 0, ""
    hp = int(b.region_hap_off[g]) + int(best)
    hr = int(b.region_hap_off[g]) + ref_hap[g]
    hap = b.hap_bases[int(b.hap_off[hp]):int(b.hap_off[hp + 1])]
    ref = b.hap_bases[int(b.hap_off[hr]):int(b.hap_off[hr + 1])]
    read = b.read_bases[int(b.read_off[r]):int(b.read_off[r + 1])]
    cig, off = oracle.sw_align(hap, read, BEST, "SoftClip")
    try:
        res = oracle.create_read_aligned_to_ref(cig, off, hap_cigars[hp], hap_starts[hp], ref_start[g], ref, read, orig_cigars[r])
    except CigarError as e:
        return e.code, 0, ""
    return (1, 0, "") if res is None else (0, res[0], res[1])
