"""Struct-of-arrays batch of assembly regions -- the host layout the C ABI (include/phmm.h) takes.

One region = the (reads, haplotypes) of one `compute_read_likelihoods` call of the reference
(src/pair_hmm/pair_hmm_likelihood_calculation_engine.rs:195-242).  A read carries the five u8
arrays of the reference's `ReadDataHolder` (src/pair_hmm/pair_hmm.rs:720-745).
"""
import numpy as np


class Read:
    """bases + base quals + insertion GOP + deletion GOP + gap continuation penalty, all len R."""
    __slots__ = ("bases", "quals", "ins", "dele", "gcp")

    def __init__(self, bases, quals, ins, dele, gcp):
        def arr(a):
            if isinstance(a, (bytes, bytearray, str)):
                a = a.encode() if isinstance(a, str) else a
                return np.frombuffer(bytes(a), dtype=np.uint8)
            return np.ascontiguousarray(a, dtype=np.uint8)
        self.bases, self.quals, self.ins, self.dele, self.gcp = arr(bases), arr(quals), arr(ins), arr(dele), arr(gcp)
        n = len(self.bases)
        # the reference asserts these (pair_hmm.rs:425-440)
        if len(self.quals) != n:
            raise ValueError("Read bases and read quals aren't the same size")
        if len(self.ins) != n:
            raise ValueError("Read bases and insertion gcp aren't the same size")
        if len(self.dele) != n:
            raise ValueError("Read bases and deletion gcp aren't the same size")
        if len(self.gcp) != n:
            raise ValueError("Read bases and overal GCP aren't the same size")

    def __len__(self):
        return len(self.bases)


class RegionBatch:
    """Flattened regions.  Arrays are numpy, C-contiguous, exactly the ABI's dtypes."""

    FIELDS = ("region_read_off", "region_hap_off", "read_off", "hap_off", "out_off",
              "read_bases", "base_q", "ins_q", "del_q", "gcp", "hap_bases")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])

    @staticmethod
    def from_regions(regions):
        """regions: iterable of (reads: list[Read], haplotypes: list[bytes|ndarray])."""
        rro, rho, ro, ho, oo = [0], [0], [0], [0], [0]
        rb, bq, iq, dq, gc, hb = [], [], [], [], [], []
        for reads, haps in regions:
            for r in reads:
                rb.append(r.bases); bq.append(r.quals); iq.append(r.ins); dq.append(r.dele); gc.append(r.gcp)
                ro.append(ro[-1] + len(r))
            for h in haps:
                h = np.frombuffer(bytes(h), dtype=np.uint8) if isinstance(h, (bytes, bytearray)) else \
                    np.ascontiguousarray(h, dtype=np.uint8)
                hb.append(h)
                ho.append(ho[-1] + len(h))
            rro.append(rro[-1] + len(reads))
            rho.append(rho[-1] + len(haps))
            oo.append(oo[-1] + len(reads) * len(haps))

        def cat(xs):
            return np.ascontiguousarray(np.concatenate(xs), dtype=np.uint8) if xs else np.zeros(0, np.uint8)
        return RegionBatch(
            region_read_off=np.asarray(rro, np.uint32), region_hap_off=np.asarray(rho, np.uint32),
            read_off=np.asarray(ro, np.uint32), hap_off=np.asarray(ho, np.uint32), out_off=np.asarray(oo, np.uint64),
            read_bases=cat(rb), base_q=cat(bq), ins_q=cat(iq), del_q=cat(dq), gcp=cat(gc), hap_bases=cat(hb))

    @staticmethod
    def concat(batches):
        """Regions of several batches, in order, as one batch (offsets shifted; out_off gaps are preserved)."""
        batches = list(batches)
        if not batches:
            return RegionBatch.from_regions([])

        def offs(name, dtype):
            parts, base = [np.zeros(1, np.int64)], 0
            for b in batches:
                a = getattr(b, name).astype(np.int64)
                parts.append(a[1:] + base)
                base += int(a[-1])
            return np.concatenate(parts).astype(dtype)

        def cat(name):
            return np.ascontiguousarray(np.concatenate([getattr(b, name) for b in batches]), dtype=np.uint8)
        return RegionBatch(region_read_off=offs("region_read_off", np.uint32), region_hap_off=offs("region_hap_off", np.uint32),
                           read_off=offs("read_off", np.uint32), hap_off=offs("hap_off", np.uint32),
                           out_off=offs("out_off", np.uint64), read_bases=cat("read_bases"), base_q=cat("base_q"),
                           ins_q=cat("ins_q"), del_q=cat("del_q"), gcp=cat("gcp"), hap_bases=cat("hap_bases"))

    # ---- shape queries ----
    @property
    def n_regions(self):
        return len(self.region_read_off) - 1

    @property
    def n_reads(self):
        return int(self.region_read_off[-1])

    @property
    def n_haps(self):
        return int(self.region_hap_off[-1])

    @property
    def n_out(self):
        return int(self.out_off[-1])

    def cells(self):
        """sum over regions of (sum of read lengths) * (sum of haplotype lengths)  (BASELINE.md)."""
        rl = np.diff(self.read_off.astype(np.int64))
        hl = np.diff(self.hap_off.astype(np.int64))
        rs = np.concatenate([[0], np.cumsum(rl)])
        hs = np.concatenate([[0], np.cumsum(hl)])
        sr = rs[self.region_read_off[1:].astype(np.int64)] - rs[self.region_read_off[:-1].astype(np.int64)]
        sh = hs[self.region_hap_off[1:].astype(np.int64)] - hs[self.region_hap_off[:-1].astype(np.int64)]
        return int(np.sum(sr * sh))

    def algorithmic_bytes(self):
        """5*sum(R) + sum(H) + 8*Nr*Nh per region  (SURVEY.md 8d)."""
        nr = np.diff(self.region_read_off.astype(np.int64))
        nh = np.diff(self.region_hap_off.astype(np.int64))
        return int(5 * int(self.read_off[-1]) + int(self.hap_off[-1]) + 8 * np.sum(nr * nh))

    def as_dict(self):
        return {f: getattr(self, f) for f in self.FIELDS}

    def region_slice(self, lo, hi):
        """Sub-batch of regions [lo, hi) with offsets rebased to 0 (used to shard across ranks)."""
        r0, r1 = int(self.region_read_off[lo]), int(self.region_read_off[hi])
        h0, h1 = int(self.region_hap_off[lo]), int(self.region_hap_off[hi])
        b0, b1 = int(self.read_off[r0]), int(self.read_off[r1])
        c0, c1 = int(self.hap_off[h0]), int(self.hap_off[h1])
        return RegionBatch(
            region_read_off=(self.region_read_off[lo:hi + 1] - np.uint32(r0)).astype(np.uint32),
            region_hap_off=(self.region_hap_off[lo:hi + 1] - np.uint32(h0)).astype(np.uint32),
            read_off=(self.read_off[r0:r1 + 1] - np.uint32(b0)).astype(np.uint32),
            hap_off=(self.hap_off[h0:h1 + 1] - np.uint32(c0)).astype(np.uint32),
            out_off=(self.out_off[lo:hi + 1] - self.out_off[lo]).astype(np.uint64),
            read_bases=self.read_bases[b0:b1], base_q=self.base_q[b0:b1], ins_q=self.ins_q[b0:b1],
            del_q=self.del_q[b0:b1], gcp=self.gcp[b0:b1], hap_bases=self.hap_bases[c0:c1])
