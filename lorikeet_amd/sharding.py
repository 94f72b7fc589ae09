"""Multi-GPU: shard whole assembly regions across ranks, no data-path collective.

Every (read, haplotype) pair and every region is independent (the reference already runs one rayon
task per region, src/assembly/assembly_region_walker.rs:210-273), so the path shards by region:
the host assigns regions to devices balanced by cells(region) with greedy longest-processing-time
(SURVEY.md 8e), each rank runs its own engine on its own GPU and results land in disjoint slices
of the output.  The only cross-rank step is optional result collection on the host
(`gather_results`), which moves the per-pair outputs, never DP state.
"""
import heapq

import numpy as np

from .batch import RegionBatch


def region_cells(batch: RegionBatch):
    rl = np.concatenate([[0], np.cumsum(np.diff(batch.read_off.astype(np.int64)))])
    hl = np.concatenate([[0], np.cumsum(np.diff(batch.hap_off.astype(np.int64)))])
    sr = rl[batch.region_read_off[1:].astype(np.int64)] - rl[batch.region_read_off[:-1].astype(np.int64)]
    sh = hl[batch.region_hap_off[1:].astype(np.int64)] - hl[batch.region_hap_off[:-1].astype(np.int64)]
    return sr * sh


def assign_regions(cells, world_size):
    """Greedy LPT: heaviest region first onto the least loaded rank.  Returns a list (per rank) of
    ascending region indices.  Deterministic (ties broken by rank, then region index)."""
    order = sorted(range(len(cells)), key=lambda g: (-int(cells[g]), g))
    heap = [(0, r) for r in range(world_size)]
    heapq.heapify(heap)
    owned = [[] for _ in range(world_size)]
    for g in order:
        load, r = heapq.heappop(heap)
        owned[r].append(g)
        heapq.heappush(heap, (load + int(cells[g]), r))
    return [sorted(o) for o in owned]


def split_contiguous(cells, n_parts):
    """Boundaries b[0..n_parts] of contiguous, cell-balanced ranges of regions: part k owns [b[k], b[k+1]).  Boundary k
    is the prefix position closest to k/n_parts of the total (ties to the left), kept monotone.  The same rule as
    phmm_split_regions (libphmm.so); no gather is needed to hand a contiguous range to an engine, so this is what
    phmm_compute_multi and bench.py's strong-scaling rows use unless the set is too heavy-tailed for it."""
    cells = [int(c) for c in cells]
    n = len(cells)
    prefix = [0]
    for c in cells:
        prefix.append(prefix[-1] + c)
    total = prefix[-1]
    bounds = [0]
    g = 0
    for k in range(1, n_parts):
        # first g with prefix[g] * n_parts >= total * k
        while g < n and prefix[g] * n_parts < total * k:
            g += 1
        if g > bounds[-1] and g > 0 and (total * k - prefix[g - 1] * n_parts) <= (prefix[g] * n_parts - total * k):
            g -= 1
        g = max(g, bounds[-1])
        bounds.append(g)
    bounds.append(n)
    return bounds


def imbalance(cells, bounds=None, owned=None):
    """Heaviest part / mean part load (1.0 = perfectly balanced) for contiguous `bounds` or LPT `owned` lists."""
    cells = [int(c) for c in cells]
    if bounds is not None:
        loads = [sum(cells[bounds[k]:bounds[k + 1]]) for k in range(len(bounds) - 1)]
    else:
        loads = [sum(cells[g] for g in o) for o in owned]
    mean = sum(loads) / max(len(loads), 1)
    return (max(loads) / mean) if mean > 0 else 1.0


def take_regions(batch: RegionBatch, regions):
    """Sub-batch made of the listed regions (any order), offsets rebased."""
    from .batch import Read
    out = []
    for g in regions:
        r0, r1 = int(batch.region_read_off[g]), int(batch.region_read_off[g + 1])
        h0, h1 = int(batch.region_hap_off[g]), int(batch.region_hap_off[g + 1])
        reads = []
        for r in range(r0, r1):
            s, e = int(batch.read_off[r]), int(batch.read_off[r + 1])
            reads.append(Read(batch.read_bases[s:e], batch.base_q[s:e], batch.ins_q[s:e], batch.del_q[s:e],
                              batch.gcp[s:e]))
        haps = [batch.hap_bases[int(batch.hap_off[a]):int(batch.hap_off[a + 1])] for a in range(h0, h1)]
        out.append((reads, haps))
    return RegionBatch.from_regions(out)


def scatter_results(batch: RegionBatch, regions, local_out, global_out):
    """Write a rank's results (regions in the order given to take_regions) into the job-wide array."""
    pos = 0
    for g in regions:
        n = int(batch.out_off[g + 1] - batch.out_off[g])
        global_out[int(batch.out_off[g]):int(batch.out_off[g]) + n] = local_out[pos:pos + n]
        pos += n
    return global_out


def compute_sharded(batch: RegionBatch, rank, world_size, compute_fn, gather=True):
    """Run `compute_fn(sub_batch) -> float64 array` on this rank's share of `batch`.

    With gather=True the per-rank outputs are collected on every rank through torch.distributed
    (all_gather_object on the default group -- host-side, result data only) and the job-wide output
    array is returned; otherwise (regions, local_out)."""
    owned = assign_regions(region_cells(batch), world_size)
    mine = owned[rank]
    local = compute_fn(take_regions(batch, mine)) if mine else np.zeros(0, np.float64)
    if not gather:
        return mine, local
    import torch.distributed as dist
    parts = [None] * world_size
    dist.all_gather_object(parts, (mine, local))
    out = np.full(batch.n_out, np.nan, dtype=np.float64)
    for regions, vals in parts:
        scatter_results(batch, regions, vals, out)
    return out
