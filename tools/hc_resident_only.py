"""The 2000 HaplotypeCaller-shaped regions as one resident batch, REPS passes (for a kernel trace). usage: hc_resident_only.py [N] [REPS] [name=value ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
regions = synth.make_regions(n, seed=5)
cells = sum(r.cells for r in regions)
hmm = HipLoglessPairHMM(0)
for kv in sys.argv[3:]:
    name, value = kv.split("=")
    hmm.set_option(name, float(value))
whole = hmm.upload(regions)
whole.compute(True)
t0 = time.perf_counter()
for _ in range(reps):
    whole.compute(True)
w = (time.perf_counter() - t0) / reps
print(f"{w * 1e3:.3f} ms per pass, {cells / w / 1e9:.0f} GCUPS, pairs {sum(r.num_reads * r.num_haps for r in regions)}, cells {cells:.4e}")
hmm.close()
