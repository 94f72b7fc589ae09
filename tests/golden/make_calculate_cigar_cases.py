"""Builds tests/golden/calculate_cigar_cases.json from the reference's own test file tests/cigar_utils_unit_tests.rs (run in the
build container, where /root/reference exists): every `test_compute_cigar(s1, s2, expected)` call of
make_test_compute_cigar_data (:34-283) -- CigarUtils::calculate_cigar(s1, s2, OverhangStrategy::InDel, NEW_SW_PARAMETERS) must
give `expected` (:21-32).  Only the data is extracted (three string literals per case, with the line they come from)."""
import json
import os
import re

SRC = "/root/reference/tests/cigar_utils_unit_tests.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "calculate_cigar_cases.json")


def main():
    text = open(SRC).read()
    cases = []
    for m in re.finditer(r'test_compute_cigar\(\s*"([^"]*)",\s*"([^"]*)",\s*"([^"]*)",?\s*\)', text):
        cases.append({"line": text.count("\n", 0, m.start()) + 1, "reference": m.group(1), "alternate": m.group(2), "expected_cigar": m.group(3)})
    assert len(cases) == text.count("test_compute_cigar(") - 1, (len(cases), text.count("test_compute_cigar("))  # (one is the definition)
    json.dump({"source": "tests/cigar_utils_unit_tests.rs:34-283", "strategy": "InDel", "parameters": [200, -150, -260, -11], "cases": cases},
              open(OUT, "w"), indent=0)
    print(len(cases), "cases ->", OUT)


if __name__ == "__main__":
    main()
