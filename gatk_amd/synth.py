"""Seeded synthetic PairHMM batches shaped like HaplotypeCaller's.

Haplotypes are a random reference window plus a few planted SNPs/indels (assembly output
shares long prefixes); reads are substrings of a random haplotype with substitution errors
drawn at the per-base Phred rate, optional indel errors, and ~0.1 % 'N' bases.  Base quals
are either "illumina" (clipped normal around Q30) or "uniform" (0..49, the distribution of
the reference's random-read generator, sam/ReadUtils.java:768-774).
"""
import numpy as np

from .batch import Batch, pack

DEFAULT_SEED = 47382911  # the reference's global RNG seed (Utils.java:51-55)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_haplotypes(rng, num_haps, hap_len, max_events=5, len_jitter=10):
    ref = _ACGT[rng.integers(0, 4, size=hap_len + 2 * len_jitter + 8)]
    haps = [ref[:hap_len].copy()]
    for _ in range(num_haps - 1):
        h = list(ref[:hap_len])
        for _e in range(int(rng.integers(0, max_events + 1))):
            pos = int(rng.integers(1, max(2, len(h) - 1)))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                h[pos] = _ACGT[(int(np.where(_ACGT == h[pos])[0][0]) + int(rng.integers(1, 4))) % 4]
            elif kind == 1 and len(h) < hap_len + len_jitter:
                h[pos:pos] = list(_ACGT[rng.integers(0, 4, size=int(rng.integers(1, 4)))])
            elif kind == 2 and len(h) > hap_len - len_jitter + 3:
                del h[pos:pos + int(rng.integers(1, 4))]
        haps.append(np.asarray(h, dtype=np.uint8))
    return haps


def make_reads(rng, haps, num_reads, read_len, qual_model="illumina", indel_rate=0.0, mean_qual=30.0,
               gop_model="const", n_rate=0.001, len_jitter=0):
    reads = []
    for _ in range(num_reads):
        h = haps[int(rng.integers(0, len(haps)))]
        rl = read_len if len_jitter == 0 else max(1, read_len - int(rng.integers(0, len_jitter + 1)))
        rl = min(rl, len(h))
        start = int(rng.integers(0, len(h) - rl + 1))
        src = h[start:start + rl + (8 if indel_rate > 0 else 0)]
        if indel_rate > 0:
            out = []
            i = 0
            while i < len(src) and len(out) < rl:
                u = rng.random()
                if u < indel_rate / 2:
                    out.append(_ACGT[int(rng.integers(0, 4))])  # insertion in the read
                elif u < indel_rate:
                    i += 1  # deletion from the read
                else:
                    out.append(src[i]); i += 1
            bases = np.asarray(out[:rl], dtype=np.uint8)
            if len(bases) == 0:
                bases = src[:1].copy()
        else:
            bases = src[:rl].copy()
        n = len(bases)
        if qual_model == "uniform":
            quals = rng.integers(0, 50, size=n).astype(np.uint8)
        else:
            quals = np.clip(np.rint(rng.normal(mean_qual, 8.0, size=n)), 6, 41).astype(np.uint8)
        err = rng.random(n) < np.power(10.0, -quals.astype(np.float64) / 10.0)
        sub = _ACGT[rng.integers(0, 4, size=n)]
        bases = np.where(err, sub, bases).astype(np.uint8)
        bases[rng.random(n) < n_rate] = ord("N")
        if gop_model == "const":
            ins = dele = None
        else:
            ins = rng.integers(20, 46, size=n).astype(np.uint8)
            dele = rng.integers(20, 46, size=n).astype(np.uint8)
        reads.append((bases, quals, ins, dele, None))
    return reads


def make_batch(num_reads, num_haps, read_len, hap_len, seed=DEFAULT_SEED, **kw) -> Batch:
    rng = np.random.default_rng(seed)
    hap_kw = {k: kw.pop(k) for k in ("max_events",) if k in kw}
    hap_jitter = kw.pop("hap_len_jitter", 10)
    haps = make_haplotypes(rng, num_haps, hap_len, len_jitter=hap_jitter, **hap_kw)
    return pack(make_reads(rng, haps, num_reads, read_len, **kw), haps)


# The named workloads (BASELINE.json "configs")
def config_c1(seed=DEFAULT_SEED):
    """128 reads (101 bp) x 16 haplotypes (~400 bp)."""
    return make_batch(128, 16, 101, 400, seed=seed)


def config_c3(seed=DEFAULT_SEED, num_reads=4096, num_haps=64):
    """Large batch: 4096 reads x 64 haplotypes of 151 bp / 600 bp (all haplotypes exactly 600 bp
    so that the cell count is the quoted 4096*64*151*600)."""
    return make_batch(num_reads, num_haps, 151, 600, seed=seed, hap_len_jitter=0, max_events=0 if False else 5)


def config_c5(seed=DEFAULT_SEED, num_reads=1024, num_haps=32):
    """Long-read / indel stress: 250 bp reads x 1000 bp haplotypes, 2-5 % indels, mean Q20."""
    return make_batch(num_reads, num_haps, 250, 1000, seed=seed, indel_rate=0.04, mean_qual=20.0, gop_model="random")


def make_regions(num_regions, seed=DEFAULT_SEED, coverage=30, read_len=151, read_len_jitter=30):
    """HaplotypeCaller-shaped stream of active regions: region span 50-300 bp plus 100 bp padding
    either side (haplotype length 250-500), 4-32 haplotypes, reads = coverage * span / read_len."""
    rng = np.random.default_rng(seed)
    regions = []
    for _ in range(num_regions):
        span = int(rng.integers(50, 301))
        hap_len = span + 200
        nh = int(rng.choice([2, 4, 6, 8, 12, 16, 24, 32], p=[.2, .25, .15, .15, .1, .08, .04, .03]))
        nr = max(4, int(round(coverage * (span + read_len) / read_len * rng.uniform(0.6, 1.4))))
        haps = make_haplotypes(rng, nh, hap_len, len_jitter=10)
        rl = min(read_len, min(len(h) for h in haps))
        regions.append(pack(make_reads(rng, haps, nr, rl, len_jitter=read_len_jitter), haps))
    return regions
