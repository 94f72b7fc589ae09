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
# the base-quality distribution as a 65 536-entry lookup table indexed by a uniform u16 (numpy's choice(p=...) costs
# 25 s of the 32 s the 10 000-region set used to take)
_QLUT = _QV[np.minimum(np.searchsorted(np.cumsum(_QP), (np.arange(65536) + 0.5) / 65536.0), len(_QV) - 1)]
_FLIP_THR = np.round(np.power(10.0, -np.arange(256) / 10.0) * 2.0 ** 32).clip(0, 2 ** 32 - 1).astype(np.uint32)  # eps(q) as a u32 threshold
_DELTA = (1 + np.arange(65536) % 3).astype(np.int8)
_GOP = np.where(np.arange(65536) < 6554, 30 + np.minimum(np.arange(65536) * 10 // 6554, 9), 40).astype(np.uint8)
_QCAP = np.where(np.arange(256) < 18, 6, np.arange(256)).astype(np.uint8)  # the engine's cap rule (engine.rs:440-444)


def make_regions(n_regions, n_reads, n_haps, hap_len, read_lens, seed, chunk=128, only=None, lengths_only=False):
    """read_lens: int or sequence of ints (each read draws its length uniformly from it).

    Every chunk of `chunk` regions has a generator of its own, seeded (seed, chunk index), so any part of a set can be
    made without the rest: `only=(lo, hi)` returns just regions [lo, hi) of the set (bit-identical to slicing the whole
    set), `lengths_only=True` returns the (n_regions, n_reads) matrix of read lengths alone (cheap: what a rank needs
    to balance a shared set by cells before it generates its own share)."""
    read_lens = np.atleast_1d(np.asarray(read_lens, dtype=np.int64))
    rmax = int(read_lens.max())
    assert rmax <= hap_len or True
    parts = {k: [] for k in ("read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")}
    lens_all = []
    sel_lo, sel_hi = (0, n_regions) if only is None else (int(only[0]), int(only[1]))
    assert 0 <= sel_lo <= sel_hi <= n_regions
    first = None
    for g0 in range(0, n_regions, chunk):
        G = min(chunk, n_regions - g0)
        if not lengths_only and (g0 + G <= sel_lo or g0 >= sel_hi or sel_lo == sel_hi):
            continue
        if first is None:
            first = g0
        rng = np.random.Generator(np.random.Philox(np.random.SeedSequence([int(seed), g0 // chunk])))
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
        if lengths_only:
            lens_all.append(rl)
            continue
        hidx = rng.integers(0, n_haps, size=(G, n_reads))
        span = np.maximum(hap_len - rl + 1, 1)
        start = np.floor(rng.random((G, n_reads)) * span).astype(np.int32)
        shape = (G, n_reads, rmax)
        # everything per base is drawn as small integers and mapped through lookup tables (4 M bases per chunk)
        col = np.minimum(start[..., None] + np.arange(rmax, dtype=np.int32)[None, None, :], hap_len - 1)
        col += ((np.arange(G, dtype=np.int32)[:, None] * n_haps + hidx.astype(np.int32)) * hap_len)[..., None]
        bases = haps.reshape(-1).take(col)
        q0 = _QLUT[rng.integers(0, 65536, size=shape, dtype=np.uint16)]
        flip = rng.integers(0, 1 << 32, size=shape, dtype=np.uint32) < _FLIP_THR[q0]   # probability eps(q)
        u = rng.integers(0, 65536, size=shape, dtype=np.uint16)
        bases = np.where(flip, (bases + _DELTA[u]) & 3, bases)   # a different base, uniformly
        q = _QCAP[q0]
        iq = _GOP[rng.integers(0, 65536, size=shape, dtype=np.uint16)]   # 10 %: uniform in 30..39, otherwise 40
        dq = _GOP[rng.integers(0, 65536, size=shape, dtype=np.uint16)]
        keep = np.arange(rmax)[None, None, :] < rl[..., None]
        parts["read_bases"].append(_ACGT[bases[keep]])
        parts["base_q"].append(q[keep])
        parts["ins_q"].append(iq[keep])
        parts["del_q"].append(dq[keep])
        parts["gcp"].append(np.full(int(keep.sum()), 10, np.uint8))
        parts["hap_bases"].append(_ACGT[haps.reshape(-1)])
        lens_all.append(rl.reshape(-1))
    if lengths_only:
        return np.concatenate(lens_all, axis=0) if lens_all else np.zeros((0, n_reads), np.int64)
    if first is None:  # empty selection
        first, lens_all = sel_lo, [np.zeros(0, np.int64)]
    n_made = min(n_regions, ((sel_hi + chunk - 1) // chunk) * chunk) - first if sel_hi > sel_lo else 0
    lens = np.concatenate(lens_all)
    arrays = {k: np.ascontiguousarray(np.concatenate(v)) if v else np.zeros(0, np.uint8) for k, v in parts.items()}
    read_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    hap_off = (np.arange(n_made * n_haps + 1, dtype=np.int64) * hap_len).astype(np.uint32)
    made = RegionBatch(
        region_read_off=(np.arange(n_made + 1, dtype=np.int64) * n_reads).astype(np.uint32),
        region_hap_off=(np.arange(n_made + 1, dtype=np.int64) * n_haps).astype(np.uint32),
        read_off=read_off, hap_off=hap_off,
        out_off=(np.arange(n_made + 1, dtype=np.int64) * n_reads * n_haps).astype(np.uint64), **arrays)
    if n_made == 0 or (sel_lo - first == 0 and sel_hi - first == n_made):
        return made
    return made.region_slice(sel_lo - first, sel_hi - first)


# The BASELINE.json configurations.
CONFIGS = {  # name -> (reads, haplotypes, haplotype length, read lengths, default regions, default seed, chunk)
    "config2": (128, 8, 300, 150, 1, 1, 128),
    "config3": (128, 8, 300, [100, 150, 250], 10000, 20250928, 128),
    "config5": (512, 64, 400, 150, 256, 7000, 8),
}


def config(name, n_regions=None, seed=None, **kw):
    nr, nh, hl, rls, n0, s0, chunk = CONFIGS[name]
    return make_regions(n0 if n_regions is None else n_regions, nr, nh, hl, rls, s0 if seed is None else seed, chunk=chunk, **kw)


def config_cells(name, n_regions=None, seed=None):
    """cells(region) for every region of a configuration, without generating it."""
    nr, nh, hl, rls, n0, s0, chunk = CONFIGS[name]
    lens = make_regions(n0 if n_regions is None else n_regions, nr, nh, hl, rls, s0 if seed is None else seed, chunk=chunk,
                        lengths_only=True)
    return lens.sum(axis=1).astype(np.int64) * (nh * hl)


def config2(n_regions=1, seed=1, **kw):
    """128 reads x 8 haplotypes, R=150, H=300 (config 2; n_regions>1 = the batched form)."""
    return config("config2", n_regions, seed, **kw)


def config3(n_regions=10000, seed=20250928, **kw):
    """128 x 8, H=300, read lengths mixed {100,150,250} inside each region (config 3/4)."""
    return config("config3", n_regions, seed, **kw)


def config5(n_regions=256, seed=7000, **kw):
    """Stress: 512 reads x 64 haplotypes, R=150, H=400 (config 5)."""
    return config("config5", n_regions, seed, **kw)


def ragged(n_regions=1536, seed=4242):
    """A long-tailed mix of regions, the way real assembly regions arrive (hard part iii;
    reference flags src/cli.rs:1568-1591,1692-1696, region loop src/assembly/assembly_region_walker.rs:210-273):
    reads per region log-normal around 60 (3 ... 5 000), haplotypes log-normal around 5 (1 ... 128), haplotype length
    60 ... 500, read lengths 30 ... 250 mixed inside a region, half of the alternative haplotypes a few bases shorter
    than the root, and about one region in 25 with a run of 'N' in one haplotype (the wildcard path).  Every region has
    a generator of its own (seed, region index)."""
    parts = []
    for g in range(n_regions):
        rng = np.random.Generator(np.random.Philox(np.random.SeedSequence([int(seed), g, 1])))
        nr = int(np.clip(np.exp(rng.normal(np.log(60.0), 1.3)), 3, 5000))
        nh = int(np.clip(np.exp(rng.normal(np.log(5.0), 0.9)), 1, 128))
        H = int(rng.integers(60, 501))
        b = make_regions(1, nr, nh, H, np.arange(30, min(250, H) + 1), seed=int(seed) * 1000003 + g)
        haps = [b.hap_bases[a * H:(a + 1) * H].copy() for a in range(nh)]
        for a in range(1, nh):
            if rng.random() < 0.5:
                haps[a] = haps[a][:H - int(rng.integers(1, 7))]
        if rng.random() < 0.04:
            a = int(rng.integers(0, nh))
            n = int(rng.integers(1, 9))
            pos = int(rng.integers(0, len(haps[a]) - n + 1))
            haps[a][pos:pos + n] = ord("N")
        b.hap_off = np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.uint32)
        b.hap_bases = np.ascontiguousarray(np.concatenate(haps))
        parts.append(b)
    return RegionBatch.concat(parts)
