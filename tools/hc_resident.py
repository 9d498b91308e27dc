"""Where the time of the HaplotypeCaller-shaped call goes: the same 2000 regions (synth.make_regions, seed 5) as one
resident batch (kernels only), as resident batches of about PART pairs each enqueued back to back on one stream
(kernels only, cut like the overlapped call), and as one compute_regions call from host arrays.
usage: hc_resident.py [N] [REPS] [PART]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
part = int(sys.argv[3]) if len(sys.argv) > 3 else 98304
regions = synth.make_regions(n, seed=5)
cells = sum(r.cells for r in regions)
hmm = HipLoglessPairHMM(0)


def timed(fn):
    for _ in range(3):
        fn()
    walls = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        walls.append(time.perf_counter() - t0)
    return float(np.median(walls)), min(walls)


def report(label, w, best):
    print(f"{label:44s} median {w * 1e3:.3f} ms (best {best * 1e3:.3f}) -> {cells / w / 1e9:.0f} GCUPS ({cells * 12 / w / 157.3e12:.3f} of the fp32 roof)", flush=True)


whole = hmm.upload(regions)
report("one resident batch, kernels only", *timed(lambda: whole.compute(True)))
st = hmm.stats()
print(f"    lane rows used {st['lane_rows_used'] / st['lane_rows_allocated']:.3f}")
chunks, cur, acc = [], [], 0
for r in regions:
    cur.append(r)
    acc += r.num_reads * r.num_haps
    if acc >= part:
        chunks.append(cur)
        cur, acc = [], 0
if cur:
    chunks.append(cur)
residents = [hmm.upload(c) for c in chunks]


def chain():
    for r in residents[:-1]:
        r.compute(False)
    residents[-1].compute(True)


report(f"{len(residents)} resident batches on one stream", *timed(chain))
call = hmm.prepare_regions(regions)
report("compute_regions from host arrays", *timed(call.compute))
st = hmm.stats()
print(f"    plan {st['host_plan_ms']:.2f} ms, lane rows used {st['lane_rows_used'] / st['lane_rows_allocated']:.3f}")
hmm.close()
