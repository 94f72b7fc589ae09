"""phmm_batch_share_prefixes on the MI355X: regions whose haplotypes share the front of their first haplotype are re-planned
(the trunk's wave parks a column, the sharers' waves sweep their suffixes from it) -- what the reference's scalar arm gets
from find_first_position_where_haplotypes_differ (pair_hmm.rs:452-464, 706-717).  Every cell is computed by the same
operations in the same order, so the results must be BIT-IDENTICAL to the unshared plan, and equal to the oracle."""
import numpy as np
import pytest
import torch

from lorikeet_amd import HipPairHMMEngine, synthetic
from lorikeet_amd.batch import RegionBatch
from oracle import oracle

pytestmark = pytest.mark.gpu


def _run(eng, batch, share):
    dev = torch.device("cuda:0")
    plan = eng.plan(batch)
    executed = plan.share_prefixes() if share else plan.cells
    tens = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    out = torch.full((batch.n_out,), float("nan"), dtype=torch.float64, device=dev)
    plan.bind_torch(tens, out)
    plan.launch()
    torch.cuda.synchronize()
    plan.status()
    res = out.cpu().numpy()
    cells, launches = plan.cells, plan.num_launches
    plan.close()
    return res, executed, cells, launches


@pytest.mark.parametrize("name", ["config2", "config5", "ragged", "short_haps"])
def test_shared_plan_is_bit_identical(hip_engine, name):
    b = {"config2": lambda: synthetic.make_regions(192, 128, 8, 300, [150], seed=77),
         "config5": lambda: synthetic.config("config5", only=(0, 6)),
         "ragged": lambda: synthetic.ragged(700, seed=99),
         "short_haps": lambda: synthetic.make_regions(300, 40, 9, 70, [30, 50, 66], seed=5)}[name]()
    plain, _, cells, _ = _run(hip_engine, b, False)
    shared, executed, cells2, launches = _run(hip_engine, b, True)
    assert cells == cells2 and np.array_equal(plain, shared)
    if name in ("config2", "config5"):
        assert executed < 0.95 * cells and launches >= 2     # (the plan did change)
    sub = b.region_slice(0, 2)
    want = oracle.compute_batch(sub.as_dict(), n_threads=4)
    assert np.max(np.abs(shared[:sub.n_out] - want)) <= 1e-9


def test_wildcards_unscalable_reads_and_repeated_calls(hip_engine):
    """A region with an 'N' haplotype keeps its plain plan; reads with gcp == 0 / base quality 0 inside a re-planned region send
    its suffix pairs to the exact pass (the general sweep needs whole pairs): same numbers to 1e-12; the call is idempotent."""
    b = synthetic.make_regions(384, 128, 8, 300, [100, 150], seed=31)
    hb = b.hap_bases.copy()
    hb[int(b.hap_off[int(b.region_hap_off[3]) + 2]) + 40] = ord("N")      # region 3: a wildcard in one haplotype
    gcp, bq = b.gcp.copy(), b.base_q.copy()
    r = int(b.region_read_off[5])
    gcp[int(b.read_off[r]) + 7] = 0                                         # region 5: a read that cannot be pre-scaled
    bq[int(b.read_off[r + 1]) + 3] = 0
    d = b.as_dict()
    d.update(hap_bases=hb, gcp=gcp, base_q=bq)
    b2 = RegionBatch(**d)
    plain, _, cells, _ = _run(hip_engine, b2, False)
    shared, executed, _, _ = _run(hip_engine, b2, True)
    assert executed < cells
    assert np.max(np.abs(plain - shared)) <= 1e-12 and np.all(np.isfinite(shared))
    touched = np.flatnonzero(plain != shared)
    lo, hi = int(b.out_off[5]), int(b.out_off[6])
    assert np.all((touched >= lo) & (touched < hi))                         # only the region with the odd reads may differ at all
    plan = hip_engine.plan(b2)
    first = plan.share_prefixes()
    assert plan.share_prefixes() == first                                   # (a second call changes nothing)
    plan.close()
    f32 = HipPairHMMEngine(0, f32_first=True)                               # no such kernels for the f32 sweep: a no-op
    p32 = f32.plan(b)
    assert p32.share_prefixes() == p32.cells
    p32.close()
    f32.close()
