"""The 2000 HaplotypeCaller-shaped regions as resident batches of about PART pairs, enqueued back to back on one stream (for a kernel trace).
usage: hc_chain_only.py [N] [REPS] [PART] [name=value ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
part = int(sys.argv[3]) if len(sys.argv) > 3 else 98304
regions = synth.make_regions(n, seed=5)
cells = sum(r.cells for r in regions)
hmm = HipLoglessPairHMM(0)
for kv in sys.argv[4:]:
    name, value = kv.split("=")
    hmm.set_option(name, float(value))
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


chain()
t0 = time.perf_counter()
for _ in range(reps):
    chain()
w = (time.perf_counter() - t0) / reps
print(f"{len(residents)} batches: {w * 1e3:.3f} ms per chain, {cells / w / 1e9:.0f} GCUPS")
hmm.close()
