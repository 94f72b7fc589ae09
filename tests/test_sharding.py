"""Multi-GPU path: regions sharded across ranks by cell count, no data-path collective.
Covered on CPU with world_size-2 gloo; the per-rank compute function here is the CPU oracle (test
infrastructure) so the sharding / scatter logic is what is under test."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from lorikeet_amd import sharding, synthetic
from lorikeet_amd.batch import Read, RegionBatch


def _ragged_batch():
    rng = np.random.default_rng(42)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    regions = []
    for g in range(11):
        nr, nh = int(rng.integers(0, 7)), int(rng.integers(1, 5))
        haps = [alpha[rng.integers(0, 4, int(rng.integers(20, 90)))] for _ in range(nh)]
        reads = []
        for _ in range(nr):
            n = int(rng.integers(5, 40))
            reads.append(Read(alpha[rng.integers(0, 4, n)], rng.integers(6, 41, n), rng.integers(30, 46, n),
                              rng.integers(30, 46, n), np.full(n, 10)))
        regions.append((reads, haps))
    return RegionBatch.from_regions(regions)


def test_lpt_assignment_is_balanced_and_complete():
    cells = np.array([100, 1, 1, 1, 50, 50, 7, 0, 3])
    owned = sharding.assign_regions(cells, 2)
    assert sorted(owned[0] + owned[1]) == list(range(len(cells)))
    loads = [int(cells[o].sum()) for o in owned]
    assert abs(loads[0] - loads[1]) <= 13 and max(loads) <= 113
    assert sharding.assign_regions(cells, 2) == owned  # deterministic
    uniform = sharding.assign_regions(np.full(16, 5), 8)
    assert all(len(o) == 2 for o in uniform)


def test_take_and_scatter_round_trip():
    from oracle import oracle
    b = _ragged_batch()
    want = oracle.compute_batch(b.as_dict())
    out = np.full(b.n_out, np.nan)
    for regions in sharding.assign_regions(sharding.region_cells(b), 3):
        sub = sharding.take_regions(b, regions)
        sharding.scatter_results(b, regions, oracle.compute_batch(sub.as_dict()), out)
    assert np.array_equal(out, want)
    assert int(sharding.region_cells(b).sum()) == b.cells()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        b = _ragged_batch()
        out = sharding.compute_sharded(b, rank, world, lambda sub: oracle.compute_batch(sub.as_dict()))
        mine, _ = sharding.compute_sharded(b, rank, world, lambda sub: oracle.compute_batch(sub.as_dict()), gather=False)
        q.put((rank, out, mine))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_matches_single_process():
    from oracle import oracle
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = oracle.compute_batch(_ragged_batch().as_dict())
    owned = set()
    for rank, out, mine in got:
        assert np.array_equal(out, want)  # every rank ends with the job-wide result
        assert not owned & set(mine)
        owned |= set(mine)
    assert owned == set(range(11))


def test_bench_shards_are_disjoint_per_rank():
    # bench.py gives rank r the regions of seed base+r: different ranks must not repeat work
    a = synthetic.config2(2, seed=1000)
    b = synthetic.config2(2, seed=1001)
    assert not np.array_equal(a.hap_bases, b.hap_bases)


def test_c_abi_assignment_matches_the_python_sharding():
    """phmm_assign_regions (what phmm_compute_multi shards by, one process over several devices) is the same greedy
    longest-processing-time assignment as the multi-process path's: same cells, same tie-breaking.  Host only."""
    from lorikeet_amd.engine import assign_regions
    for batch in (_ragged_batch(), synthetic.config3(37, seed=3), synthetic.config2(5, seed=1)):
        cells = sharding.region_cells(batch)
        for parts in (1, 2, 3, 8, 64):
            want = np.zeros(batch.n_regions, np.uint32)
            for p, regions in enumerate(sharding.assign_regions(cells, parts)):
                want[regions] = p
            assert np.array_equal(assign_regions(batch, parts), want), parts


def test_contiguous_split_c_abi_matches_python_and_is_balanced():
    """phmm_split_regions == sharding.split_contiguous (what phmm_compute_multi and bench.py's strong-scaling rows shard
    by): same boundaries, every region in exactly one part, and on the uniform sets of BASELINE.json the heaviest part
    is within a region of the mean.  Host only."""
    from lorikeet_amd.engine import split_regions
    for batch in (_ragged_batch(), synthetic.config3(37, seed=3), synthetic.config2(5, seed=1), synthetic.ragged(40, seed=2)):
        cells = sharding.region_cells(batch)
        for parts in (1, 2, 3, 8, 64):
            want = sharding.split_contiguous(cells, parts)
            got = split_regions(batch, parts)
            assert got.tolist() == want, (parts, got, want)
            assert want[0] == 0 and want[-1] == batch.n_regions and all(a <= b for a, b in zip(want, want[1:]))
    cells = synthetic.config_cells("config3")
    for parts in (2, 4, 8):
        b = sharding.split_contiguous(cells, parts)
        assert sharding.imbalance(cells, bounds=b) < 1.001
    c5 = synthetic.config_cells("config5")
    assert sharding.imbalance(c5, bounds=sharding.split_contiguous(c5, 8)) == 1.0


def test_strong_scaling_shards_cover_the_fixed_set_exactly_once():
    """bench.py's config3_10k / config5_256 rows: rank r generates only regions [b[r], b[r+1]) of the shared set; together the
    ranks hold the whole set, bit-identical to generating it in one piece."""
    full = synthetic.config3(300)
    cells = synthetic.config_cells("config3", 300)
    assert np.array_equal(cells, sharding.region_cells(full))
    for world in (2, 3):
        b = sharding.split_contiguous(cells, world)
        parts = [synthetic.config3(300, only=(b[r], b[r + 1])) for r in range(world)]
        whole = RegionBatch.concat(parts)
        for f in RegionBatch.FIELDS:
            assert np.array_equal(getattr(whole, f), getattr(full, f)), f


def test_every_rank_of_an_8_gpu_node_still_fills_its_chip():
    """VERDICT r2, multi-GPU readiness: no 8-GPU node exists to measure on, so the plan is checked where it can be -- on the
    host.  Rank r's share of BASELINE.json configs[3] at N = 8 (1 250 of the 10 000 regions) and of configs[4] (32 of the
    256 stress regions, 512 x 64, H = 400) must still take the chained kernel for every cell, with enough work items (one
    wave each) to put two waves on each of the chip's 1 024 SIMDs several times over, and runs of at least four reads
    (shorter runs pay the lane pipeline's fill too often).  phmm_plan_describe: the planner without a device."""
    from lorikeet_amd.engine import plan_describe
    for name, world in (("config3", 8), ("config5", 8), ("config3", 4), ("config5", 2)):
        cells = synthetic.config_cells(name)
        bounds = sharding.split_contiguous(cells, world)
        for rank in (0, world - 1):
            shard = synthetic.config(name, only=(bounds[rank], bounds[rank + 1]))
            info = plan_describe(shard)
            assert info.cells == int(cells[bounds[rank]:bounds[rank + 1]].sum())
            assert info.chain_cells == info.cells, (name, world, rank)                  # nothing falls back to the per-read kernel
            assert info.chain_items >= 4 * 2 * 1024, (name, world, rank, info.chain_items)  # >= 4 rounds of 2 waves per SIMD
            assert info.min_reads_per_run >= 4 and info.n_chain_launches == 1
            assert info.dominant_kernel.decode().startswith("phmm_forward_chain_k<16,"), info.dominant_kernel
    # the whole set on one GPU is the N = 1 row of the same experiment
    one = plan_describe(synthetic.config("config5"))
    assert one.chain_cells == one.cells == int(synthetic.config_cells("config5").sum())


def test_the_plan_says_what_it_sweeps_and_where_the_padding_is():
    """phmm_plan_info.swept_cells (= phmm_batch_executed_cells, the bench's rows.*.executed_per_cell): steps x 64 lanes x K columns
    of every wave, with the columns beyond a haplotype's end and the empty haplotype slots counted apart.  The uniform 128 x 8 batch
    (H = 300 in 16 lanes x 19 columns = 304): 4 / 300 of the useful cells in column padding exactly, no empty slot (eight haplotypes,
    four per wave), a few per cent of steps without a read row; the ragged mix pays for its shapes (tools/ragged_padding.py,
    NOTEBOOK.md 20.6)."""
    from lorikeet_amd.engine import plan_describe
    u = plan_describe(synthetic.config2(1024))
    assert u.pad_column_cells * 300 == u.cells * 4 and u.pad_slot_cells == 0
    steps = u.swept_cells - u.cells - u.pad_column_cells - u.pad_slot_cells
    assert 0.01 * u.cells < steps < 0.04 * u.cells            # SUM / RESET rows (2 per 150) + the fill of each run
    r = plan_describe(synthetic.ragged())
    assert r.swept_cells >= r.cells + r.pad_column_cells + r.pad_slot_cells
    assert 1.08 < r.swept_cells / r.cells < 1.20 and r.pad_column_cells > 0 and r.pad_slot_cells > 0
    one = plan_describe(synthetic.config2(1))                  # a lone region: 64 lanes x 5 columns per pair, one read per wave
    assert one.pad_column_cells * 300 == one.cells * 20 and one.swept_cells > one.cells

