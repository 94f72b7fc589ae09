"""phmm_compute_multi: one process, several engines (one per device on a multi-GPU node; several on the one device of the
test box), whole regions sharded by cells with greedy LPT and nothing exchanged between them."""
import numpy as np
import pytest

from lorikeet_amd import HipPairHMMEngine, PhmmError, synthetic
from lorikeet_amd import _lib
from lorikeet_amd.batch import Read, RegionBatch
from lorikeet_amd.engine import assign_regions, compute_multi
from oracle import oracle
from test_sharding import _ragged_batch

pytestmark = pytest.mark.gpu


def test_several_engines_one_call():
    engines = [HipPairHMMEngine(0) for _ in range(3)]
    ragged = _ragged_batch()            # regions without reads, 1-4 haplotypes, very different cell counts
    want = oracle.compute_batch(ragged.as_dict(), n_threads=4)
    got = compute_multi(engines, ragged)
    assert np.max(np.abs(got - want)) <= 1e-9
    for b in (synthetic.config3(40, seed=8), synthetic.config2(2, seed=9)):   # the second: fewer regions than engines
        single = engines[0].compute(b)
        assert np.max(np.abs(compute_multi(engines, b) - single)) <= 1e-12
        assert np.max(np.abs(compute_multi(engines[:1], b) - single)) == 0.0   # one engine: plain phmm_compute
    part = assign_regions(ragged, 3)
    assert set(part.tolist()) <= {0, 1, 2} and len(part) == ragged.n_regions
    for e in engines:
        e.close()


def test_errors_of_a_share_reach_the_caller():
    engines = [HipPairHMMEngine(0) for _ in range(2)]
    hap = np.full(40, ord("A"), np.uint8)
    n = 8
    weird = ([Read(hap[:n].copy(), np.full(n, 93), np.zeros(n, int), np.zeros(n, int), np.full(n, 10))], [hap])
    plain = ([Read(hap[:20].copy(), np.full(20, 30), np.full(20, 40), np.full(20, 40), np.full(20, 10))], [hap])
    b = RegionBatch.from_regions([weird, plain, plain, plain])
    with pytest.raises(PhmmError) as e:
        compute_multi(engines, b)
    assert e.value.code == _lib.PHMM_ERR_POSITIVE_RESULT and "greater than 0.0" in str(e.value)
    good = synthetic.config2(3, seed=4)
    assert np.max(np.abs(compute_multi(engines, good) - engines[0].compute(good))) <= 1e-12   # the handles stay usable
    bad = synthetic.make_regions(2, 4, 2, 100, 50, seed=1)
    bad.read_off = bad.read_off.copy()
    bad.read_off[2] = bad.read_off[1] - 1
    with pytest.raises(PhmmError) as e:
        compute_multi(engines, bad)
    assert e.value.code == _lib.PHMM_ERR_INVALID_ARG and "monotonic" in str(e.value)
    for x in engines:
        x.close()
