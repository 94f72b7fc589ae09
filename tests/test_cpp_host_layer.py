"""The C++ host layer above the C ABI (lorikeet_amd/csrc/host/lorikeet_pair_hmm.hpp) mirrors the reference's
Rust surface; tests/cpp/reference_tests.cpp restates the reference's own tests against it.  The binary is built by
`make -C lorikeet_amd/csrc` (__graft_entry__.build())."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

EXE = os.path.join(ROOT, "tests", "cpp", "reference_tests")


def test_host_layer_binary_is_built_and_has_no_cpu_fallback():
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    from lorikeet_amd import _lib
    if _lib.load().phmm_device_count() > 0:
        pytest.skip("a HIP device is present")
    r = subprocess.run([EXE, os.path.join(GOLDEN, "pairhmm-testdata.txt")], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


def test_host_layer_mirrors_the_reference_surface():
    hpp = open(os.path.join(ROOT, "lorikeet_amd", "csrc", "host", "lorikeet_pair_hmm.hpp")).read()
    for name in ("class PairHMM", "compute_log10_likelihoods", "get_log_likelihood_array", "do_not_use_tristate_correction",
                 "class PairHMMLikelihoodCalculationEngine", "compute_read_likelihoods", "class PairHMMInputScoreImputator",
                 "ins_open_penalties", "gap_continuation_penalties", "enum class PCRErrorModel", "enum class AVXMode",
                 "class AlleleLikelihoods", "class AssemblyResultSet", "class SmithWatermanAligner",
                 "enum class OverhangStrategy", "struct SmithWatermanAlignmentResult", "NEW_SW_PARAMETERS", "struct BestAllele",
                 "best_alleles_breaking_ties_main", "haplotype_alignment_tiebreaking_priority", "reference_tiebreaking_priority",
                 "struct AlignmentUtils", "create_read_aligned_to_ref", "struct CigarUtils", "calculate_cigar"):
        assert name in hpp, name


@pytest.mark.gpu
def test_reference_tests_in_cpp_pass_on_the_gpu():
    r = subprocess.run([EXE, os.path.join(GOLDEN, "pairhmm-testdata.txt")], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "17 tests, 0 failed" in r.stdout
    for name in ("test_likelihoods_avx", "make_basic_likelihood_tests", "test_compute_likelihoods",
                 "make_haplotype_indexing_provider", "make_big_read_hmm_provider", "rayon_worker_pattern", "region_pipeline_worker_pattern",
                 "smith_waterman_asserted_cases", "test_for_identical_alignments_with_differing_flank_lengths", "test_best_alleles", "make_read_aligned_to_ref_data", "make_test_compute_cigar_data"):
        assert "PASS " + name in r.stdout
