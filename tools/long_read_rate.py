"""Throughput of the long-read pass (reads of more than 1022 bases): cells per second of one call of such reads.

usage: python tools/long_read_rate.py [read_len] [num_reads] [num_haps] [ordinary_reads]

ordinary_reads: that many 151-base reads in the same region, scored by the fp32 pass while the long-read pass runs
beside it; the call is then also timed without the long reads.
"""
import sys
import time

import numpy as np

from gatk_amd import pairhmm
from gatk_amd.batch import pack


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    haps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    ordinary = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, n + 200)]
    recs = []
    for _ in range(reads):
        read = hap[50:50 + n].copy()
        read[::97] = acgt[rng.integers(0, 4, len(read[::97]))]
        recs.append((read, rng.integers(20, 40, n), None, None, None))
    short = []
    for _ in range(ordinary):
        at = int(rng.integers(0, n))
        read = hap[at:at + 151].copy()
        read[::31] = acgt[rng.integers(0, 4, len(read[::31]))]
        short.append((read, rng.integers(20, 40, len(read)), None, None, None))
    hap_list = []
    for a in range(haps):
        x = hap.copy()
        x[100 + a] = acgt[(np.searchsorted(acgt, x[100 + a]) + 1) % 4]
        hap_list.append(x)
    b = pack(recs + short, hap_list)
    h = pairhmm.HipLoglessPairHMM(0)
    try:
        if short:
            only = pack(short, hap_list)
            h.compute_batch(only)
            t0 = time.perf_counter()
            for _ in range(5):
                h.compute_batch(only)
            print(f"{ordinary} reads of 151 bases alone: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
        h.compute_batch(b)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out = h.compute_batch(b)
        dt = (time.perf_counter() - t0) / reps
        cells = reads * haps * n * (n + 200)
        print(f"{reads} reads of {n} bases x {haps} haplotypes of {n + 200}: {dt * 1e3:.1f} ms per call, {cells / dt / 1e9:.1f} GCUPS, "
              f"finite {np.isfinite(out).all()}")
    finally:
        h.close()


if __name__ == "__main__":
    main()
