"""Synthetic assembly regions of the shapes BASELINE.json / SURVEY.md 8(d) name.

Per region: one root haplotype of length H (uniform ACGT); the other Nh-1 haplotypes are the root
with 1-3 SNVs.  Each read copies R bases from a random haplotype at a random start and flips each
base with probability eps(q).  Base quals are i.i.d. from {37:.60, 32:.15, 27:.10, 22:.08, 12:.05,
6:.02} and then pass the engine's cap rule (q < 18 -> 6; ...engine.rs:440-444 with the default
threshold, cli.rs:1859-1863); ins/del quals are 40 with 10 % of positions uniform in 30..39 (Q45
capped by the PCR model); gcp = 10 (cli.rs:1484-1488).  Deterministic for a given seed (numpy
Philox), vectorised over regions so the 10k-region sets build in seconds.
"""
import numpy as np

from .batch import RegionBatch

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_QV = np.array([37, 32, 27, 22, 12, 6], dtype=np.uint8)
_QP = np.array([0.60, 0.15, 0.10, 0.08, 0.05, 0.02])


def make_regions(n_regions, n_reads, n_haps, hap_len, read_lens, seed, chunk=128):
    """read_lens: int or sequence of ints (each read draws its length uniformly from it)."""
    read_lens = np.atleast_1d(np.asarray(read_lens, dtype=np.int64))
    rmax = int(read_lens.max())
    assert rmax <= hap_len or True
    rng = np.random.Generator(np.random.Philox(int(seed)))
    parts = {k: [] for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    lens_all = []
    for g0 in range(0, n_regions, chunk):
        G = min(chunk, n_regions - g0)
        root = rng.integers(0, 4, size=(G, 1, hap_len), dtype=np.int8)
        haps = np.repeat(root, n_haps, axis=1)
        if n_haps > 1:
            nsnv = rng.integers(1, 4, size=(G, n_haps))
            gi, hi = np.meshgrid(np.arange(G), np.arange(n_haps), indexing="ij")
            for s in range(3):
                pos = rng.integers(0, hap_len, size=(G, n_haps))
                delta = rng.integers(1, 4, size=(G, n_haps), dtype=np.int8)
                m = (s < nsnv) & (hi > 0)
                haps[gi[m], hi[m], pos[m]] = (haps[gi[m], hi[m], pos[m]] + delta[m]) & 3
        rl = read_lens[rng.integers(0, len(read_lens), size=(G, n_reads))]
        hidx = rng.integers(0, n_haps, size=(G, n_reads))
        span = np.maximum(hap_len - rl + 1, 1)
        start = np.floor(rng.random((G, n_reads)) * span).astype(np.int64)
        col = np.minimum(start[..., None] + np.arange(rmax)[None, None, :], hap_len - 1)
        bases = haps[np.arange(G)[:, None, None], hidx[..., None], col]
        q = _QV[rng.choice(len(_QV), size=(G, n_reads, rmax), p=_QP)]
        flip = rng.random((G, n_reads, rmax)) < np.power(10.0, -q.astype(np.float64) / 10.0)
        delta = rng.integers(1, 4, size=(G, n_reads, rmax), dtype=np.int8)
        bases = np.where(flip, (bases + delta) & 3, bases)
        q = np.where(q < 18, 6, q).astype(np.uint8)
        low = rng.random((G, n_reads, rmax)) < 0.10
        iq = np.where(low, rng.integers(30, 40, size=(G, n_reads, rmax)), 40).astype(np.uint8)
        low = rng.random((G, n_reads, rmax)) < 0.10
        dq = np.where(low, rng.integers(30, 40, size=(G, n_reads, rmax)), 40).astype(np.uint8)
        keep = np.arange(rmax)[None, None, :] < rl[..., None]
        parts["read_bases"].append(_ACGT[bases[keep]])
        parts["base_q"].append(q[keep])
        parts["ins_q"].append(iq[keep])
        parts["del_q"].append(dq[keep])
        parts["gcp"].append(np.full(int(keep.sum()), 10, np.uint8))
        parts["hap_bases"].append(_ACGT[haps.reshape(-1)])
        lens_all.append(rl.reshape(-1))
    lens = np.concatenate(lens_all)
    arrays = {k: np.ascontiguousarray(np.concatenate(v)) for k, v in parts.items()}
    read_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    hap_off = (np.arange(n_regions * n_haps + 1, dtype=np.int64) * hap_len).astype(np.uint32)
    return RegionBatch(
        region_read_off=(np.arange(n_regions + 1, dtype=np.int64) * n_reads).astype(np.uint32),
        region_hap_off=(np.arange(n_regions + 1, dtype=np.int64) * n_haps).astype(np.uint32),
        read_off=read_off, hap_off=hap_off,
        out_off=(np.arange(n_regions + 1, dtype=np.int64) * n_reads * n_haps).astype(np.uint64), **arrays)


# The BASELINE.json configurations.
def config2(n_regions=1, seed=1):
    """128 reads x 8 haplotypes, R=150, H=300 (config 2; n_regions>1 = the batched form)."""
    return make_regions(n_regions, 128, 8, 300, 150, seed)


def config3(n_regions=10000, seed=20250928):
    """128 x 8, H=300, read lengths mixed {100,150,250} inside each region (config 3/4)."""
    return make_regions(n_regions, 128, 8, 300, [100, 150, 250], seed)


def config5(n_regions=256, seed=7000):
    """Stress: 512 reads x 64 haplotypes, R=150, H=400 (config 5)."""
    return make_regions(n_regions, 512, 64, 400, 150, seed, chunk=8)
