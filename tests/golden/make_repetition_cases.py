"""Builds tests/golden/repetition_cases.json and tests/golden/consolidate_cigar_cases.json from the reference's own tests
(run in the build container, where /root/reference exists):

  * every assertion of `test_find_number_of_repetitions` and `test_find_number_of_repetitions_full_array`
    (tests/variant_context_utils_unit_tests.rs:23-294): the arguments of VariantContextUtils::find_number_of_repetitions /
    find_number_of_repetitions_main and the asserted count -- what pins oracle/engine_oracle.c's restatement of the two
    functions the PCR indel model's tandem-repeat scan is made of (engine.rs:528-611);
  * every assertion of `test_consolidate_cigar` (tests/haplotype_unit_tests.rs:96-146): haplotype CIGAR, pad size and the
    asserted result of Haplotype::get_consolidated_padded_cigar -- what pins the first step of the projection
    (alignment_utils.rs:84-100).
Only data is extracted (string literals, integers, booleans), by pattern; no code of the reference is copied."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
VCU = "/root/reference/tests/variant_context_utils_unit_tests.rs"
HAP = "/root/reference/tests/haplotype_unit_tests.rs"

S = r'"([A-Za-z]*)"\.as_bytes\(\)'
SHORT = re.compile(r"find_number_of_repetitions\(\s*%s,\s*%s,\s*(true|false),?\s*\),\s*(\d+)\s*\)" % (S, S))
MAIN = re.compile(r"find_number_of_repetitions_main\(\s*%s,\s*(\d+),\s*(\d+),\s*%s,\s*(\d+),\s*(\d+),\s*(true|false),?\s*\),\s*(\d+)\s*\)" % (S, S))
CONS = re.compile(r'make_hcf_for_cigar\("([ACGT]+)",\s*"([0-9MIDNSHPX=]+)"\)\s*\.get_consolidated_padded_cigar\((\d+)\)\s*'
                  r'\.unwrap_or_else\(\|_\|\s*panic!\("[^"]*"\)\),\s*CigarString::try_from\("([0-9MIDNSHPX=]+)"\)')


def line_of(text, pos):
    return text.count("\n", 0, pos) + 1


def main():
    text = open(VCU).read()
    short = [{"source": "tests/variant_context_utils_unit_tests.rs:%d" % line_of(text, m.start()), "repeat_unit": m.group(1),
              "test_string": m.group(2), "leading_repeats": m.group(3) == "true", "expected": int(m.group(4))} for m in SHORT.finditer(text)]
    full = [{"source": "tests/variant_context_utils_unit_tests.rs:%d" % line_of(text, m.start()), "repeat_unit_full": m.group(1),
             "offset_in_repeat_unit_full": int(m.group(2)), "repeat_unit_length": int(m.group(3)), "test_string_full": m.group(4),
             "offset_in_test_string_full": int(m.group(5)), "test_string_length": int(m.group(6)), "leading_repeats": m.group(7) == "true",
             "expected": int(m.group(8))} for m in MAIN.finditer(text)]
    # nothing may be missed: as many cases as the file has calls inside assert_eq!
    assert len(short) == len(re.findall(r"VariantContextUtils::find_number_of_repetitions\(", text)), len(short)
    assert len(full) == len(re.findall(r"VariantContextUtils::find_number_of_repetitions_main\(", text)), len(full)
    json.dump({"find_number_of_repetitions": short, "find_number_of_repetitions_main": full},
              open(os.path.join(HERE, "repetition_cases.json"), "w"), indent=1)
    htext = open(HAP).read()
    cons = [{"source": "tests/haplotype_unit_tests.rs:%d" % line_of(htext, m.start()), "bases": m.group(1), "cigar": m.group(2),
             "pad_size": int(m.group(3)), "expected": m.group(4)} for m in CONS.finditer(htext)]
    assert len(cons) == len(re.findall(r"\.get_consolidated_padded_cigar\(", htext)), len(cons)
    json.dump({"get_consolidated_padded_cigar": cons}, open(os.path.join(HERE, "consolidate_cigar_cases.json"), "w"), indent=1)
    print("%d + %d repetition cases, %d consolidate cases" % (len(short), len(full), len(cons)))


if __name__ == "__main__":
    main()
