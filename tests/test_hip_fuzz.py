"""Property-based parity: hypothesis draws ragged regions (tiny to moderate reads and haplotypes, arbitrary bytes
incl. 'N', qualities over the range the engine can produce) and the HIP path must agree with the oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from lorikeet_amd.batch import Read, RegionBatch
from oracle import oracle

pytestmark = pytest.mark.gpu

bases = st.sampled_from(list(b"ACGTN"))
qual = st.integers(0, 93)
indel_qual = st.integers(6, 93)  # >= MIN_USABLE_Q_SCORE as cap_minimum_read_qualities guarantees


@st.composite
def read_st(draw):
    n = draw(st.integers(0, 70))
    return Read(bytes(draw(st.lists(bases, min_size=n, max_size=n))), draw(st.lists(qual, min_size=n, max_size=n)),
                draw(st.lists(indel_qual, min_size=n, max_size=n)), draw(st.lists(indel_qual, min_size=n, max_size=n)),
                draw(st.lists(st.integers(1, 60), min_size=n, max_size=n)))


@st.composite
def region_st(draw):
    reads = draw(st.lists(read_st(), min_size=0, max_size=6))
    haps = draw(st.lists(st.lists(bases, min_size=1, max_size=90).map(bytes), min_size=1, max_size=6))
    return reads, haps


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(regions=st.lists(region_st(), min_size=1, max_size=5), force_l=st.sampled_from([None, "16", "32", "64"]),
       chain=st.sampled_from([None, "3"]), streams=st.sampled_from([None, "1", "2", "4"]))
def test_hip_equals_oracle_on_drawn_batches(regions, force_l, chain, streams):
    import os
    from lorikeet_amd import HipPairHMMEngine
    b = RegionBatch.from_regions(regions)
    want = oracle.compute_batch(b.as_dict())
    eng = HipPairHMMEngine(0)
    for k, v in (("force_L", force_l), ("force_chain", chain), ("force_streams", streams)):
        if v is not None:
            eng.set_switch(k, int(v))
    got = eng.compute(b)
    eng.close()
    assert got.shape == want.shape
    inf = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf)
    if (~inf).any():
        assert float(np.max(np.abs(got[~inf] - want[~inf]))) <= 1e-9
