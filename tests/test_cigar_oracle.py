"""oracle/cigar_oracle.c restates the CIGAR algebra behind create_read_aligned_to_ref (src/reads/cigar_builder.rs,
src/reads/alignment_utils.rs).  Pinned here by the DATA of the reference's own tests, restated case by case:
tests/cigar_builder_unit_tests.rs and tests/alignment_utils_unit_tests.rs (line numbers at each test)."""
import itertools

import numpy as np
import pytest

from oracle import oracle
from oracle.oracle import CigarError


# ---- tests/cigar_builder_unit_tests.rs ------------------------------------------------------------------------------------
def test_cigar_algebra():  # :25-45
    leading = [[], ["10H"], ["10S"], ["10H", "10S"]]
    middle = [["10M"], ["10M", "10I", "10M"], ["10M", "10D", "10M"]]
    trailing = [[], ["10H"], ["10S"], ["10S", "10H"]]
    for a, b, c in itertools.product(leading, middle, trailing):
        assert oracle.cigar_builder(a + b + c)[0] == "".join(a + b + c)


@pytest.mark.parametrize("elements,expected", [  # :63-72
    (["10M", "10D"], "10M"), (["10D", "10M"], "10M"), (["10H", "10D", "10M"], "10H10M"), (["10S", "10D", "10M"], "10S10M"),
    (["10S", "10D", "10M", "10S"], "10S10M10S"), (["10M", "10D", "10S"], "10M10S"), (["10M", "10D", "10H"], "10M10H"),
    (["10S", "10M", "10D", "10H"], "10S10M10H")])
def test_initial_and_final_deletion(elements, expected):
    assert oracle.cigar_builder(elements)[0] == expected


@pytest.mark.parametrize("elements,expected", [  # :90-99
    (["10M", "10D"], "10M10D"), (["10D", "10M"], "10D10M"), (["10H", "10D", "10M"], "10H10D10M"), (["10S", "10D", "10M"], "10S10D10M"),
    (["10S", "10D", "10M", "10S"], "10S10D10M10S"), (["10M", "10D", "10S"], "10M10D10S"), (["10M", "10D", "10H"], "10M10D10H"),
    (["10S", "10M", "10D", "10H"], "10S10M10D10H")])
def test_retain_deletions(elements, expected):
    assert oracle.cigar_builder(elements, remove_deletions_at_ends=False)[0] == expected


@pytest.mark.parametrize("elements,expected", [  # merge_consecutive :115-128, tricky :130-132, indel_sandwich :134-155
    (["10H", "10H", "10M"], "20H10M"), (["10S", "10M", "10M"], "10S20M"), (["10S", "10M", "10S", "10S"], "10S10M20S"),
    (["10S", "10M", "10I", "10I", "10I", "10S", "10H"], "10S10M30I10S10H"),
    (["10S", "10S", "10M", "10M", "10I", "10I", "10S", "10H"], "20S20M20I10S10H"),
    (["10H", "10H", "10D", "10D", "10M"], "20H10M"),
    (["10M", "10I", "10D", "10M"], "10M10D10I10M"), (["10M", "10D", "10I", "10M"], "10M10D10I10M"),
    (["10M", "10I", "10D", "10I", "10M"], "10M10D20I10M"), (["10M", "10I", "10D", "10I", "10D", "10I", "10M"], "10M20D30I10M"),
    (["10M", "10I", "10D", "10I", "10M", "10D", "10I", "10M"], "10M10D20I10M10D10I10M"),
    (["10D", "10I", "10M"], "10I10M"), (["10M", "10I", "10D"], "10M10I"), (["10M", "10D", "10I"], "10M10I"),
    (["10M", "10D", "10I", "10S"], "10M10I10S"), (["10S", "10D", "10I", "10M"], "10S10I10M"),
    (["10S", "10I", "10D", "10I", "10M"], "10S20I10M")])
def test_merge_consecutive_and_indel_sandwich(elements, expected):
    assert oracle.cigar_builder(elements)[0] == expected


@pytest.mark.parametrize("elements", [  # :164-175
    ["10S"], ["10S", "10S"], ["10S", "10D"], ["10S", "10D", "10S"], ["10S", "10D", "10D", "10S"], ["10S", "10H", "10M"],
    ["10M", "10H", "10S"], ["10M", "10H", "10M"], ["10M", "10S", "10M"]])
def test_invalid(elements):
    with pytest.raises(CigarError):
        oracle.cigar_builder(elements)


@pytest.mark.parametrize("elements,leading,trailing", [  # :199-222
    (["10M"], 0, 0), (["10S", "10M"], 0, 0), (["10M", "10S"], 0, 0), (["10M", "10I", "10D", "10M"], 0, 0),
    (["10M", "10D", "10I", "10M"], 0, 0), (["10D", "10I", "10M"], 10, 0), (["10D", "10D", "10I", "10M"], 20, 0),
    (["10D", "10D", "10I", "10D", "10M"], 30, 0), (["10S", "10D", "10D", "10I", "10D", "10M"], 30, 0),
    (["10M", "10I", "10D"], 0, 10), (["10M", "10D", "10I"], 0, 10), (["10M", "10D", "10I", "10D"], 0, 20),
    (["10M", "10D", "10I", "10D", "10S", "10H"], 0, 20),
    (["10H", "10S", "10D", "10M", "10D", "10I", "10D", "10S", "10H"], 10, 20)])
def test_removed_deletions(elements, leading, trailing):
    assert oracle.cigar_builder(elements)[1:] == (leading, trailing)


# ---- tests/alignment_utils_unit_tests.rs ----------------------------------------------------------------------------------
LEFT_ALIGN = [  # make_left_align_indel_data :614-666
    ("ACGT", "ACGT", "4M", "4M"), ("ACCT", "ACGT", "4M", "4M"), ("ACGT", "ACAT", "2M1X1M", "2M1X1M"),
    ("AAATTT", "AAACCCTTT", "3M3I3M", "3M3I3M"), ("CCCTTT", "AAACCCTTT", "3I6M", "3I6M"), ("AAACCC", "AAACCCTTT", "6M3I", "6M3I"),
    ("AAACCC", "AAACCGTTT", "6M3I", "6M3I"), ("AAACCCTTT", "AAATTT", "3M3D3M", "3M3D3M"),
    ("AAACCCTTT", "AAACCCCCCTTT", "5M3I4M", "3M3I6M"), ("AAACCCTTT", "AAACCCCCCTTT", "6M3I3M", "3M3I6M"),
    ("AAACCCTTT", "AAGCCCCCCTGT", "6M3I3M", "3M3I6M"), ("AAACGCGCGCGTTT", "AAACGCGCGCGCGCGTTT", "7M4I7M", "3M4I11M"),
    ("CCGCCG", "CCGCCGCCG", "6M3I", "3I6M"), ("ACCGCCG", "TCCGCCGCCG", "7M3I", "1M3I6M"),
    ("AAACCCCCCTTT", "AAACCCTTT", "5M3D4M", "3M3D6M"), ("AAACCCCCCTTT", "AAACCCTTT", "6M3D3M", "3M3D6M"),
    ("AAACGCGCGCGCGCGTTT", "AAACGCGCGCGTTT", "7M4D7M", "3M4D11M"),
    ("AAACCCTTTGGGAAA", "AAACCCCCCTTTGGGGGGAAA", "6M3I6M3I3M", "3M3I6M3I6M"),
    ("AAACCCTTTGGGGGGAAA", "AAACCCCCCTTTGGGAAA", "6M3I6M3D3M", "3M3I6M3D6M"),
    ("AAACCCCCTTT", "AAACCCCCTTT", "4M3I3D4M", "11M"), ("AAACCCCCTTT", "AAACCCCCTTT", "4M3D3I4M", "11M"),
    ("AAACCCCCTTT", "AAACCCCCTTT", "3M3I2M3D3M", "11M"), ("AACGCGCGCGTT", "AACGCGCGCGCGCGTT", "2M2I8M2I2M", "2M4I10M"),
    ("AACGCGCGCGCGCGTT", "AACGCGCGCGTT", "2M2D8M2D2M", "2M4D10M")]


@pytest.mark.parametrize("ref,read,cigar,expected", LEFT_ALIGN)
def test_left_align_indels_with_clips_and_reference_context(ref, read, cigar, expected):
    """test_with_clips_and_reference_context (:491-602): hard / soft clips on either side, extra reference in front / behind."""
    rng = np.random.default_rng(len(ref) * 131 + len(read))
    rnd = lambda n: bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, n))  # noqa: E731
    for lh, th, ls, ts, front, back in itertools.product((0, 5), (0, 5), (0, 5), (0, 5), (0, 10), (0, 10)):
        read_bases = rnd(ls) + read.encode() + rnd(ts)
        ref_bases = rnd(front) + ref.encode() + rnd(back)
        clip = lambda core: ("%dH" % lh if lh else "") + ("%dS" % ls if ls else "") + core + ("%dS" % ts if ts else "") + ("%dH" % th if th else "")  # noqa: E731
        got, _, _ = oracle.left_align_indels(clip(cigar), ref_bases, read_bases, front)
        assert got == clip(expected), (lh, th, ls, ts, front, back)


def _trim_by_reference(cigar, start, end, expected):  # test_trim_cigar :668-685
    if len(oracle.parse_cigar(expected)) == 1 and expected.endswith("D"):
        return  # "trimming throws error if all but deletion elements are trimmed"
    want = oracle.cigar_builder([expected])[0]
    assert oracle.trim_cigar(cigar, start, end, True)[0] == want, (cigar, start, end)


def test_trim_cigar_data():  # make_trim_cigar_data :687-784
    for op in "D=XM":
        for my_length in range(1, 6):
            for start in range(0, my_length - 1):
                for end in range(start, my_length):
                    length = end - start + 1
                    for pad_op in "DM":
                        for left_pad in range(2):
                            for right_pad in range(2):
                                cig = ("%d%s" % (left_pad, pad_op) if left_pad else "") + "%d%s" % (my_length, op) + \
                                      ("%d%s" % (right_pad, pad_op) if right_pad else "")
                                _trim_by_reference(cig, start + left_pad, end + left_pad, "%d%s" % (length, op))
    for left_pad in (0, 1, 2, 5):
        for right_pad in (0, 1, 2, 5):
            length = left_pad + right_pad
            if length > 0:
                for ins_size in (1, 10):
                    for start in range(0, left_pad + 1):
                        for stop in range(left_pad, length):
                            lrem, rrem = left_pad - start, stop - left_pad + 1
                            ins = "%dI" % ins_size
                            _trim_by_reference("%dM%s%dM" % (left_pad, ins, right_pad), start, stop,
                                               ("%dM" % lrem if lrem else "") + ins + ("%dM" % rrem if rrem else ""))
    for cig, s, e, want in [("3M2D4M", 0, 8, "3M2D4M"), ("3M2D4M", 2, 8, "1M2D4M"), ("3M2D4M", 2, 6, "1M2D2M"), ("3M2D4M", 3, 6, "2D2M"),
                            ("3M2D4M", 4, 6, "1D2M"), ("3M2D4M", 5, 6, "2M"), ("3M2D4M", 6, 6, "1M"), ("2M3I4M", 0, 5, "2M3I4M"),
                            ("2M3I4M", 1, 5, "1M3I4M"), ("2M3I4M", 1, 4, "1M3I3M"), ("2M3I4M", 2, 4, "3I3M"), ("2M3I4M", 2, 3, "3I2M"),
                            ("2M3I4M", 2, 2, "3I1M"), ("2M3I4M", 3, 4, "2M"), ("2M3I4M", 3, 3, "1M"), ("2M3I4M", 4, 4, "1M")]:
        _trim_by_reference(cig, s, e, want)


@pytest.mark.parametrize("cigar,start,end,expected", [  # make_trim_cigar_by_bases_data :799-819
    ("2M3I4M", 0, 8, "2M3I4M"), ("2M3I4M", 1, 8, "1M3I4M"), ("2M3I4M", 2, 8, "3I4M"), ("2M3I4M", 3, 8, "2I4M"), ("2M3I4M", 4, 8, "1I4M"),
    ("2M3I4M", 4, 7, "1I3M"), ("2M3I4M", 4, 6, "1I2M"), ("2M3I4M", 4, 5, "1I1M"), ("2M3I4M", 4, 4, "1I"), ("2M3I4M", 5, 5, "1M"),
    ("2M2D2I", 0, 3, "2M2I"), ("2M2D2I", 1, 3, "1M2I"), ("2M2D2I", 2, 3, "2I"), ("2M2D2I", 3, 3, "1I"), ("2M2D2I", 2, 2, "1I"),
    ("2M2D2I", 1, 2, "1M1I"), ("2M2D2I", 0, 1, "2M"), ("2M2D2I", 1, 1, "1M")])
def test_trim_cigar_by_bases(cigar, start, end, expected):
    assert oracle.trim_cigar(cigar, start, end, False)[0] == expected


@pytest.mark.parametrize("a,b,expected", [("%dM" % i, "%dM" % i, "%dM" % i) for i in range(1, 5)] + [  # :833-884
    ("3M", "2M3D1M", "2M3D1M"), ("3M1I2M", "2M1D3M", "2M1D1M1I2M"), ("1M1D2M", "1M1D3M", "1M2D2M"), ("1M2D2M", "1M1D1M1I2M", "1M2D2M"),
    ("1M1I4M", "5M", "1M1I4M"), ("1M2D2M", "5M", "1M2D2M"), ("108M14D24M2M18I29M92M1000M", "2M1I3M", "2M1I3M")])
def test_apply_cigar_to_cigar(a, b, expected):
    assert oracle.apply_cigar_to_cigar(a, b) == expected


@pytest.mark.parametrize("original,shifted,expected", [  # :902-917
    ("30M", "30M", "30M"), ("30M", "15M6I15M", "15M6I15M"), ("5S30M", "30M", "5S30M"), ("5H30M", "30M", "5H30M"),
    ("5H5S30M", "30M", "5H5S30M"), ("30M5H", "30M", "30M5H"), ("30M5S", "30M", "30M5S"), ("10H30M5S5H", "30M", "10H30M5S5H"),
    ("10H10M6D6M6D50M5S5H", "10M6I50M", "10H10M6I50M5S5H")])
def test_append_clipped_elements_from_original_cigar(original, shifted, expected):
    assert oracle.append_clipped_elements(shifted, original) == expected


@pytest.mark.parametrize("cigar,start,expected", [  # :1015-1022
    ("30M5D20M", 50, 55), ("30M5I20M", 50, 45), ("55M", 50, 50), ("30M5D30M5D30M", 80, 90), ("30M5D30M5I30M", 80, 80)])
def test_read_start_on_reference_haplotype(cigar, start, expected):
    assert oracle.read_start_on_reference_haplotype(cigar, start) == expected


# ---- create_read_aligned_to_ref (:111-398): the reference's aligner is restated by oracle.sw_align ----------------------------
BEST_HAP = [10, -15, -30, -5]  # ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS


def _aligned_to_ref(read, hap_bases, hap_cigar, hap_start, ref_bases, ref_start, original_cigar="10M"):
    cig, off = oracle.sw_align(hap_bases, read, BEST_HAP, "SoftClip")
    return oracle.create_read_aligned_to_ref(cig, off, hap_cigar, hap_start, ref_start, ref_bases, read, original_cigar)


def test_read_aligned_to_ref_data():  # make_read_aligned_to_ref_data :156-215
    hap = "ACTGAAGGTTCC"
    all_m = "%dM" % len(hap)
    for i in range(-1, len(hap)):
        read = bytearray(hap.encode())
        if i != -1:
            read[i] = ord("A")
        assert _aligned_to_ref(bytes(read), hap, all_m, 0, hap, 10) == (10, all_m)
    for pad_front in range(1, 10):
        assert _aligned_to_ref("N" * pad_front + hap, hap, all_m, 0, hap, 10) == (10, "%dI%s" % (pad_front, all_m))
    for pad_back in range(1, 10):
        assert _aligned_to_ref(hap + "N" * pad_back, hap, all_m, 0, hap, 10) == (10, "%s%dI" % (all_m, pad_back))
    for ref_start in range(1, 10):
        for hap_start in range(ref_start, 10 + ref_start):
            assert _aligned_to_ref(hap, hap, all_m, hap_start, hap, ref_start) == (ref_start + hap_start, all_m)


def test_read_aligned_to_ref_long_cases():  # :216-290
    hap = "ACTGTGGGTTCCTCTTATTTTATTTCTACATCAATGTTCATATTTAACTTATTATTTTATCTTATTTTTAAATTTCTTTTATGTTGAGCCTTGATGAAAGCCATAGGTTCTCTCATATAATTGTATGTGTATGTATGTATATGTACATAATATATACATATATGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTGTATTACATAATATATACATATATGTATATATTATGTATATGTACATAATATATACATATATG"
    read = "ATGTACATAATATATACATATATGTATATGTATGTACATAATATATACGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTACATAATATATACGTATATGTATGTGTATGTGTATTACATAATATATACATATATGTATATATTATGTATATGTACATAATAT"
    assert _aligned_to_ref(read, hap, "%dM" % len(hap), 500, hap, 10130100) == (10130740, "28M6D214M")
    reference = "CTGAACGTAACCAAAATCAATATGGATACTGAGAAATACTATTTAATAAAGACATAAATTAGACTGCTAAAAAAAATTAAAGAAATTTCAAAAGAGAATCCACCTCTTTTCCTTGCCAGTGCTCAAAAGTGAGTGTGAATCTGGTGGCTGTGGGGCTGTTTTTGGTGTGGCTCTTTGGACCAGCCTGCCTGGTAATTCAAGCCTGCCTCTCATTTCTG"
    haplotype = "CTGAACGTAACCAAAATCAATATGGATACTGAGAAATACTATTTAATAAAGACATAAATTAGACTGCTAAAAAAAATTAAAGAAATTTCAAAAGAGAATCCACCTCTTTTCCTTGCCAGTGCTCAAAAGTGAGTGTGAATCTGGTGGCTGCGGGGCTGTTTTTGGTGTGGCTCTTTGGACCAGCCTGCCTGGTAATTCAAGCCTGCCTCTCATTTCTG"
    assert _aligned_to_ref("GCTGCTTTTGGTGTGGCTCTTT", haplotype, "%dM" % len(haplotype), 575, reference, 215239171) == (215239171 + 575 + 154, "22M")
    reference = "GGGATCCTGCTACAAAGGTGAAACCCAGGAGAGTGTGGAGTCCAGAGTGTTGCCAGGACCCAGGCACAGGCATTAGTGCCCGTTGGAGAAAACAGGGGAATCCCGAAGAAATGGTGGGTCCTGGCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC"
    haplotype = "GGGATCCTGCTACAAAGGTGAAACCCAGGAGAGTGTGGAGTCCAGAGTGTTGCCAGGACCCAGGCACAGGCATTAGTGCCCGTTGGAGAAAACGGGAATCCCGAAGAAATGGTGGGTCCTGGCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC"
    read = "CCCATCCGTGAGATCTTCCCAGGGCAGCTCCCCTCTGTGGAATCCAATCTGTCTTCCATCCTGC"
    assert _aligned_to_ref(read, haplotype, "93M2D92M", 553, reference, 13011) == (13011 + 553 + 123, "64M")


def _mutate(seq, mutations):  # Mutation::apply / mutate_sequence :293-409
    n_mis = 0
    for pos, length, op in sorted(mutations, key=lambda m: m[0]):
        if op == "M":
            if pos < len(seq):
                seq = seq[:pos] + ("C" if seq[pos] == "A" else "A") + seq[pos + 1:]
        elif op == "I":
            seq = seq[:pos] + "GTCAGTTA"[:length] + seq[pos:]
        else:
            seq = seq[:pos] + seq[pos + length:]
        n_mis += length
    return seq, n_mis


def _mismatches(read, cigar, reference, pos):
    """AlignmentUtils::get_mismatch_count (src/reads/alignment_utils.rs:870-960) for a whole read: mismatching bases of the
    aligned blocks (start_on_read 0, all bases)."""
    n, r, p = 0, 0, pos
    for e in oracle.parse_cigar(cigar):
        length, op = int(e) >> 4, "MIDNSHP=X"[int(e) & 15]
        if op in "M=X":
            for k in range(length):
                if p + k < len(reference) and read[r + k] != reference[p + k]:
                    n += 1
            r += length
            p += length
        elif op in "IS":
            r += length
        elif op in "DN":
            p += length
    return n


def test_complex_read_aligned_to_ref():  # make_complex_read_aligned_to_ref :442-486
    all_mutations = [(1, 1, "M"), (2, 1, "M"), (3, 1, "I"), (7, 1, "D")]
    subsets = [list(c) for k in range(1, 4) for c in itertools.combinations(all_mutations, k)]
    reference_bases = "ACTGACTGACTG"
    padded = "NNNN" + reference_bases + "NNNN"
    checked = 0
    for muts in subsets:
        hap, hap_mis = _mutate(reference_bases, muts)
        hcig, hoff = oracle.sw_align(padded, hap, [3, -1, -4, -3], "SoftClip")  # ORIGINAL_DEFAULT
        for read_muts in subsets:
            read, read_mis = _mutate(hap, read_muts)
            pos, cigar = _aligned_to_ref(read, hap, oracle.cigar_to_string(hcig), hoff, padded, 0)
            assert _mismatches(read, cigar, padded, pos) <= hap_mis + read_mis, (hap, read, cigar, pos)
            checked += 1
    assert checked == len(subsets) ** 2


# ---- CigarUtils::calculate_cigar: tests/cigar_utils_unit_tests.rs:21-283, the data in tests/golden/calculate_cigar_cases.json -------
def test_calculate_cigar_cases_of_the_reference():
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "calculate_cigar_cases.json")))
    assert len(gold["cases"]) >= 110
    for c in gold["cases"]:
        assert oracle.calculate_cigar(c["reference"], c["alternate"], gold["parameters"], gold["strategy"]) == c["expected_cigar"], c
