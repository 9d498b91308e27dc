"""Packed form of one PairHMM.computeLikelihoods call.

The reference hands the PairHMM a list of reads, each with five byte arrays of the read's
length (bases, base quals, insertion GOP, deletion GOP, overall GCP), and the list of
haplotype alleles of the likelihood matrix (PairHMM.java:190-214).  `Batch` is the same data
concatenated into flat uint8 arrays with int32 offset tables, which is the layout the C-ABI
takes and the device keeps in HBM.
"""
from dataclasses import dataclass

import numpy as np

DEFAULT_INSERTION_DELETION_QUAL = 45  # sam/GATKSAMRecord.java:61
DEFAULT_GCP = 10  # constant gap-continuation penalty HaplotypeCaller passes for every base


@dataclass
class Batch:
    read_bases: np.ndarray    # uint8[total_read_len]
    read_quals: np.ndarray    # uint8[total_read_len]
    ins_gop: np.ndarray       # uint8[total_read_len]
    del_gop: np.ndarray       # uint8[total_read_len]
    gcp: np.ndarray           # uint8[total_read_len]
    read_offsets: np.ndarray  # int32[num_reads + 1]
    hap_bases: np.ndarray     # uint8[total_hap_len]
    hap_offsets: np.ndarray   # int32[num_haps + 1]
    read_mapq: np.ndarray = None  # uint8[num_reads] or None: cap on the read's base qualities
    # ins_gop, del_gop and gcp may be None: the engine's defaults (Q45, Q45, 10)

    @property
    def num_reads(self):
        return len(self.read_offsets) - 1

    @property
    def num_haps(self):
        return len(self.hap_offsets) - 1

    @property
    def read_lengths(self):
        return np.diff(self.read_offsets)

    @property
    def hap_lengths(self):
        return np.diff(self.hap_offsets)

    @property
    def cells(self):
        """Sum over (read, haplotype) pairs of readLen * hapLen: the numerator of GCUPS."""
        return int(self.read_lengths.astype(np.int64).sum()) * int(self.hap_lengths.astype(np.int64).sum())

    def read_slice(self, lo, hi):
        """The same haplotypes with reads lo..hi-1 only (how one large batch is split over devices)."""
        a, b = int(self.read_offsets[lo]), int(self.read_offsets[hi])
        cut = lambda x: None if x is None else x[a:b]  # the optional arrays stay optional
        return Batch(self.read_bases[a:b], self.read_quals[a:b], cut(self.ins_gop), cut(self.del_gop), cut(self.gcp),
                     (self.read_offsets[lo:hi + 1] - a).astype(np.int32), self.hap_bases, self.hap_offsets,
                     None if self.read_mapq is None else self.read_mapq[lo:hi])


def _as_u8(x):
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8)
    if isinstance(x, str):
        return np.frombuffer(x.encode("ascii"), dtype=np.uint8)
    return np.asarray(x, dtype=np.uint8)


def _concat(parts):
    offs = np.zeros(len(parts) + 1, dtype=np.int64)
    if parts:
        offs[1:] = np.cumsum([len(p) for p in parts])
    if offs[-1] >= 2 ** 31:
        raise ValueError("batch too large for int32 offsets")
    data = np.concatenate(parts) if parts and offs[-1] > 0 else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(data, dtype=np.uint8), offs.astype(np.int32)


def pack(reads, haplotypes):
    """reads: iterable of (bases, quals, ins_gop, del_gop, gcp); ins_gop/del_gop may be None
    (the read has no BI/BD tags: flat Q45, sam/GATKSAMRecord.java:298-322) and gcp may be None
    (constant 10).  haplotypes: iterable of base sequences.  Length mismatches between the five
    arrays of a read raise ValueError, as PairHMM.java:276-279 raises IllegalArgumentException."""
    rb, rq, ri, rd, rg = [], [], [], [], []
    for k, rec in enumerate(reads):
        bases, quals, ins, dele, gcp = rec
        bases = _as_u8(bases)
        n = len(bases)
        quals = _as_u8(quals)
        ins = np.full(n, DEFAULT_INSERTION_DELETION_QUAL, np.uint8) if ins is None else _as_u8(ins)
        dele = np.full(n, DEFAULT_INSERTION_DELETION_QUAL, np.uint8) if dele is None else _as_u8(dele)
        gcp = np.full(n, DEFAULT_GCP, np.uint8) if gcp is None else _as_u8(gcp)
        for name, arr in (("read quals", quals), ("read insertion quals", ins), ("read deletion quals", dele), ("overall GCP", gcp)):
            if len(arr) != n:
                raise ValueError(f"Read bases and {name} aren't the same size: {n} vs {len(arr)} (read {k})")
        rb.append(bases); rq.append(quals); ri.append(ins); rd.append(dele); rg.append(gcp)
    read_bases, read_offsets = _concat(rb)
    read_quals, _ = _concat(rq)
    ins_gop, _ = _concat(ri)
    del_gop, _ = _concat(rd)
    gcp_all, _ = _concat(rg)
    hap_bases, hap_offsets = _concat([_as_u8(h) for h in haplotypes])
    return Batch(read_bases, read_quals, ins_gop, del_gop, gcp_all, read_offsets, hap_bases, hap_offsets)
