#!/usr/bin/env python3
"""How often the engine's fp32/fp64 arithmetic could change a downstream decision.

filterPoorlyModeledReads (genotyper/ReadLikelihoods.java:1059-1093) drops a read when its best likelihood is
under  -4 * min(2, ceil(readLength * maximumErrorPerBase))  -- a hard comparison (:1090), so a read whose
best likelihood sits within the engine's error bound (1e-3 in log10) of that threshold could be kept by one
implementation and dropped by another.  This counts, on HaplotypeCaller-shaped synthetic regions, the reads
within 1e-3 of the threshold and the reads whose flag differs between the engine (device likelihoods, device
flags) and the fp64 CPU restatement (its likelihoods, its flags).  Prints one JSON line.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regions", type=int, default=1000)
    ap.add_argument("--rate", type=float, default=0.02, help="maximumErrorPerBase")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()

    regions = synth.make_regions(args.regions, seed=5)
    hmm = HipLoglessPairHMM(0)
    res = hmm.upload(regions)
    res.compute(sync=True)
    got = res.download()
    flags = res.poorlyModeledReads(args.rate)
    res.free()
    hmm.close()

    reads = near = flipped = flagged = 0
    worst = 0.0
    closest = np.inf
    for b, m, f in zip(regions, got, flags):
        if m.size == 0:
            continue
        rc, want = O.compute_likelihoods(b, O.LOGLESS, fast_threads=args.threads)
        assert rc == 0
        lens = np.diff(b.read_offsets)
        rc, want_flags = O.poorly_modelled_reads(np.ascontiguousarray(want.T), lens, args.rate)
        assert rc == 0
        fin = np.isfinite(want)
        worst = max(worst, float(np.abs(m[fin] - want[fin]).max()))
        threshold = -4.0 * np.minimum(2.0, np.ceil(lens * args.rate))
        distance = np.abs(want.max(axis=1) - threshold)
        reads += len(lens)
        near += int((distance <= 1e-3).sum())
        closest = min(closest, float(distance.min()))
        flipped += int((np.asarray(f) != np.asarray(want_flags)).sum())
        flagged += int(np.asarray(want_flags).sum())
    print(json.dumps({
        "workload": f"{args.regions} synthetic HaplotypeCaller-shaped regions (gatk_amd.synth.make_regions, seed 5)",
        "maximum_error_per_base": args.rate, "reads": reads, "poorly_modelled_reads_fp64": flagged,
        "reads_within_1e-3_of_the_threshold": near, "closest_distance_to_the_threshold": closest,
        "reads_whose_flag_differs_between_engine_and_fp64": flipped,
        "max_abs_err_log10": worst}))


if __name__ == "__main__":
    main()
