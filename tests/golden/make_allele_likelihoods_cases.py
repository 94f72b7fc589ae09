"""Builds tests/golden/allele_likelihoods_cases.json: the reference's container tests
`test_normalize_cap_worst_lk` (tests/allele_likelihoods_unit_tests.rs:725-770) and `test_filer_poorly_modeled_reads`
(:399-442 with make_good_and_bad_likelihoods :560-581) as VALUES (run in the build container, where /root/reference exists).

The reference draws its inputs from ThreadRng, so there are no literal matrices to lift.  What the tests hold is
  * the shapes: SAMPLE_SETS x ALLELE_SETS (:37-68; sample counts, allele counts, which allele is the reference) -- read here from
    the test file by pattern -- and read counts `gen_range(0, 100)` per sample (data_set_reads :915-942);
  * the inputs' law: N(0, 1) per (allele, read) for the normalisation (fill_two_with_random_likelihoods :127-152 -- positive
    values included), and for the filter zeros with every ODD read at -10000 for all alleles (:560-581), threshold -100;
  * the arguments: normalize_likelihoods(-0.001, true) (:735), filter_poorly_modeled_evidence(|_| -100.0) (:420);
  * and the EXPECTED side, computed in the test itself (not by the library): per read `best = max over all alleles`, new value
    `max(best - 0.001, old)`, a read whose best is -inf left alone (:741-766); exactly the even reads survive, in order, with
    their columns (:422-441).
This script draws seeded inputs of those shapes and that law (values on a 2^-20 grid, so the JSON is exact), evaluates the
tests' expected side, and stores both.  One column per case is set to -inf to reach the branch at :752-755, which the
reference's N(0, 1) inputs never take.  Only data and the tests' arithmetic on it: no code of the reference is copied."""
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tests/allele_likelihoods_unit_tests.rs"


def sets(text):
    body = text.split("static ref SAMPLE_SETS", 1)[1].split("static ref ALLELE_SETS", 1)
    samples = [len(re.findall(r"\d+", m)) for m in re.findall(r"vec!\[([\d,\s]+)\]", body[0].split("= vec![", 1)[1])]
    allele_block = body[1].split("const EPSILON", 1)[0]
    alleles = []
    # (the inner vec![...] groups: split on "vec![" after the outer one, flags of the ByteArrayAllele::new(.., true|false) calls in each)
    for group in allele_block.split("= vec![", 1)[1].split("vec![")[1:]:
        alleles.append([flag == "true" for flag in re.findall(r'ByteArrayAllele::new\("[ACGT]+"\.as_bytes\(\),\s*(true|false)\)', group)])
    return samples, alleles


def main():
    text = open(SRC).read()
    samples, alleles = sets(text)
    assert samples == [3, 1, 6] and [len(a) for a in alleles] == [3, 1, 2, 2, 2], (samples, alleles)
    assert "result.normalize_likelihoods(-0.001, true);" in text and "filter_poorly_modeled_evidence(Box::new(|_read| -100.0))" in text
    rng = np.random.default_rng(20260929)
    cases = []
    for si, n_samples in enumerate(samples):
        for ai, flags in enumerate(alleles):
            n_alleles = len(flags)
            ref = flags.index(True) if True in flags else -1
            per_sample = []
            for s in range(n_samples):
                n_reads = int(rng.integers(0, 41))   # (the reference: 0..100; 0..40 keeps the fixture small)
                v = np.round(rng.normal(0.0, 1.0, (n_alleles, n_reads)) * (1 << 20)) / (1 << 20)
                if n_reads > 3 and s == 0:
                    v[:, 3] = -np.inf
                # :741-766 -- the test's own expectation (the library's single-allele no-op, allele_likelihoods.rs:386-389, gives the
                # same numbers: max(v - 0.001, v) == v)
                want = v.copy()
                for r in range(n_reads):
                    best = max(v[a, r] for a in range(n_alleles))
                    if best != -np.inf:
                        for a in range(n_alleles):
                            want[a, r] = max(best - 0.001, v[a, r])
                # :560-581, :422-441
                good_bad = np.zeros((n_alleles, n_reads))
                good_bad[:, 1::2] = -10000.0
                kept = [r for r in range(n_reads) if (r & 1) == 0]
                assert len(kept) == (n_reads + 1) // 2
                enc = lambda m: [[("-inf" if x == -np.inf else repr(float(x))) for x in row] for row in m]  # noqa: E731
                per_sample.append({"n_reads": n_reads, "likelihoods": enc(v), "normalized": enc(want),
                                   "good_and_bad": enc(good_bad), "kept_reads": kept, "filtered": enc(good_bad[:, kept])})
            cases.append({"source": "tests/allele_likelihoods_unit_tests.rs:37-68 SAMPLE_SETS[%d] x ALLELE_SETS[%d]" % (si, ai),
                          "n_alleles": n_alleles, "reference_allele_index": ref, "samples": per_sample})
    out = {"normalize": {"source": "tests/allele_likelihoods_unit_tests.rs:725-770", "maximum_likelihood_difference_cap": -0.001, "symmetric": True},
           "filter": {"source": "tests/allele_likelihoods_unit_tests.rs:399-442, :560-581", "threshold": -100.0}, "cases": cases}
    path = os.path.join(HERE, "allele_likelihoods_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, len(cases), "cases,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
