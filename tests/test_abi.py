"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/phmm.h declares, refuses to run without a GPU (no fallback), and its host-built device
tables are bit-identical to the oracle's restatement of the reference tables."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from lorikeet_amd import _lib
from oracle import oracle


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "phmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phmm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 15
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(declared) == bound, (set(declared) ^ bound)
    for name in declared:
        assert getattr(lib, name) is not None


def test_header_cites_the_reference_interface():
    text = open(os.path.join(ROOT, "include", "phmm.h")).read()
    for cite in ("pair_hmm.rs:345-375", "pair_hmm.rs:217-341", "vector_pair_hmm_unit_tests.rs"):
        assert cite in text


def test_tables_match_oracle_bit_for_bit():
    lib = _lib.load()
    olib = oracle.lib()
    p = _lib.f64p()
    n = lib.phmm_table_eps(C.byref(p))
    assert n == 256
    eps = np.ctypeslib.as_array(p, shape=(n,)).copy()
    for q in range(256):
        assert eps[q] == olib.oracle_qual_to_error_prob(q)
    n = lib.phmm_table_match_to_match(C.byref(p))
    assert n == 256 * 257 // 2
    mm = np.ctypeslib.as_array(p, shape=(n,)).copy()
    for mx in range(256):
        for mn in range(mx + 1):
            assert mm[(mx * (mx + 1) >> 1) + mn] == olib.oracle_match_to_match_prob(mn, mx), (mn, mx)
    # the oracle's own 0..=254 table is a prefix of ours
    olen = olib.oracle_mm_table_len()
    otab = np.ctypeslib.as_array(olib.oracle_mm_prob_table(), shape=(olen,))
    assert np.array_equal(mm[:olen], otab)


def test_no_cpu_fallback_without_a_device():
    lib = _lib.load()
    if lib.phmm_device_count() > 0:
        pytest.skip("a HIP device is present")
    assert lib.phmm_create(0, 0) is None
    assert b"no CPU fallback" in lib.phmm_last_error(None)
    from lorikeet_amd import HipPairHMMEngine, PhmmError
    with pytest.raises(PhmmError):
        HipPairHMMEngine(0)
    from lorikeet_amd.pair_hmm import forward
    with pytest.raises(PhmmError):
        forward(b"ACGT", b"ACG", [30] * 3, [40] * 3, [40] * 3, [10] * 3)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lorikeet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("the oracle's", "").replace("against the oracle", ""), \
                    os.path.join(dirpath, f)


def test_batch_builder_layout():
    from lorikeet_amd.batch import Read, RegionBatch
    r1 = Read(b"ACGT", [30] * 4, [40] * 4, [41] * 4, [10] * 4)
    r2 = Read(b"AC", [20] * 2, [40] * 2, [41] * 2, [10] * 2)
    b = RegionBatch.from_regions([([r1, r2], [b"ACGTA", b"AC"]), ([], [b"A"]), ([r2], [b"ACG"])])
    assert b.n_regions == 3 and b.n_reads == 3 and b.n_haps == 4
    assert list(b.region_read_off) == [0, 2, 2, 3] and list(b.region_hap_off) == [0, 2, 3, 4]
    assert list(b.read_off) == [0, 4, 6, 8] and list(b.hap_off) == [0, 5, 7, 8, 11]
    assert list(b.out_off) == [0, 4, 4, 5]
    assert b.cells() == 6 * 7 + 0 + 2 * 3
    assert b.algorithmic_bytes() == 5 * 8 + 11 + 8 * (4 + 0 + 1)
    s = b.region_slice(2, 3)
    assert s.n_reads == 1 and bytes(s.hap_bases) == b"ACG" and list(s.out_off) == [0, 1]
    with pytest.raises(ValueError):
        Read(b"ACG", [30] * 2, [40] * 3, [40] * 3, [10] * 3)
