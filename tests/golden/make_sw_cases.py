"""Builds tests/golden/smith_waterman_cases.json from the reference's own test file
tests/smith_waterman_aligner_unit_tests.rs (run in the build container, where /root/reference exists):

  * the ASSERTED cases (expected alignment offset and CIGAR), transcribed below as data with the line each comes
    from; every literal sequence is checked to occur in the reference file so the transcription cannot drift;
  * the three long (reference, read) byte vectors of `test_avx_mode` (:402-997), extracted by regex.  The reference
    asserts on them that its vector arm equals its scalar arm for three parameter sets x three overhang strategies
    (:999-1103); the CIGARs listed next to them are NOT asserted there (commented out, :1028-1029) and are not used.
Only data is extracted -- no code of the reference is copied."""
import json
import os
import re

SRC = "/root/reference/tests/smith_waterman_aligner_unit_tests.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "smith_waterman_cases.json")

PARAMS = {  # smith_waterman_aligner.rs:11-26  (match, mismatch, gap open, gap extend)
    "ORIGINAL_DEFAULT": [3, -1, -4, -3], "STANDARD_NGS": [25, -50, -110, -6], "NEW_SW_PARAMETERS": [200, -150, -260, -11],
    "ALIGNMENT_TO_BEST_HAPLOTYPE_SW_PARAMETERS": [10, -15, -30, -5]}

ASSERTED = [
    # (test, line, reference, read, params, strategy, expected offset, expected cigar)
    ("make_test_read_alignment_to_ref_complex_alignment", 230, "AAAGGACTGACTG", "ACTGACTGACTG", "ORIGINAL_DEFAULT", "SoftClip", 1, "12M"),
    ("make_test_odd_no_alignment", 254, "AAAGACTACTG", "AACGGACACTG", [50, -100, -220, -12], "SoftClip", 1, "2M2I3M1D4M"),
    ("make_test_odd_no_alignment", 261, "AAAGACTACTG", "AACGGACACTG", [200, -50, -300, -22], "SoftClip", 0, "11M"),
    ("test_indels_at_start_and_end", 288, "AAACCCCC", "CCCCCGGG", "ORIGINAL_DEFAULT", "SoftClip", 3, "5M3S"),
    ("test_degenerate_alignment_with_indels_at_both_ends", 305, "TGTGTGTGTGTGTGACAGAGAGAGAGAGAGAGAGAGAGAGAGAGA",
     "ACAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGA", "STANDARD_NGS", "SoftClip", 14, "31M20S"),
    ("get_substrings_match_tests", 382, "AAACCCCC", "CCCCC", "ORIGINAL_DEFAULT", "SoftClip", 3, "5M"),
    ("get_substrings_match_tests", 383, "AAACCCCC", "CCCCC", "ORIGINAL_DEFAULT", "InDel", 0, "3D5M"),
    ("get_substrings_match_tests", 384, "AAACCCCC", "CCCCC", "ORIGINAL_DEFAULT", "LeadingInDel", 0, "3D5M"),
    ("get_substrings_match_tests", 385, "AAACCCCC", "CCCCC", "ORIGINAL_DEFAULT", "Ignore", 3, "5M"),
]
LITERALS = ["AAAGGACTGACTG", "ACTGACTGACTG", "AAAGACTACTG", "AACGGACACTG", '"AAA"', "CCCCC", '"GGG"', "2M2I3M1D4M", "5M3S", "31M20S",
            "TGTGTGTGTGTGTGACAGAGAGAGAGAGAGAGAGAGAGAGAGAGA", "ACAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGAGA", "3D5M"]


def main():
    text = open(SRC).read()
    for lit in LITERALS:
        assert lit in text, lit
    cases = []
    for test, line, ref, read, params, strategy, off, cigar in ASSERTED:
        cases.append({"source": "tests/smith_waterman_aligner_unit_tests.rs:%d (%s)" % (line, test), "reference": ref,
                      "read": read, "params": PARAMS[params] if isinstance(params, str) else params, "strategy": strategy,
                      "expected_offset": off, "expected_cigar": cigar})
    # the flank-length test (:320-378): the two padded pairs, aligned with NEW_SW_PARAMETERS / SoftClip; the reference
    # asserts a property of the result (same indel elements), kept as inputs
    flank = {}
    for name in ("padded_ref", "padded_hap", "not_padded_ref", "not_padded_hap"):
        m = re.search(r'let %s = "([ACGT-]+)"' % name, text)
        flank[name] = m.group(1).replace("-", "")
    pairs = []
    for k in (1, 2, 3):
        vec = {}
        for what in ("ref", "read"):
            m = re.search(r"let %s_%d: Vec<u8> = vec!\[(.*?)\];" % (what, k), text, flags=re.S)
            vec[what] = bytes(int(x) for x in re.findall(r"\d+", m.group(1))).decode()
        pairs.append({"source": "tests/smith_waterman_aligner_unit_tests.rs test_avx_mode pair %d" % k,
                      "reference": vec["ref"], "read": vec["read"]})
    json.dump({"params": PARAMS, "asserted": cases, "flank_pairs": flank, "avx_equals_scalar_pairs": pairs},
              open(OUT, "w"), indent=1)
    print("wrote", OUT, len(cases), "asserted cases,", [(len(p["reference"]), len(p["read"])) for p in pairs])


if __name__ == "__main__":
    main()
