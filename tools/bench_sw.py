"""Smith-Waterman throughput: haplotype-to-reference alignments of a HaplotypeCaller-like region set on the
GPU against the oracle's scalar CPU restatement (one thread)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import smithwaterman as sw  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(7)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
refs, alts = [], []
for _ in range(pairs):
    n = int(rng.integers(250, 501))
    ref = rng.choice(list(b"ACGT"), size=n).astype(np.uint8)
    alt = ref.copy()
    for _ in range(int(rng.integers(0, 6))):
        alt[rng.integers(n)] = rng.choice(list(b"ACGT"))
    p = int(rng.integers(20, n - 20))
    alt = np.concatenate([alt[:p], alt[p + int(rng.integers(0, 10)):]])
    refs.append(ref.tobytes())
    alts.append(alt.tobytes())
cells = sum(len(r) * len(a) for r, a in zip(refs, alts))
sw.align_batch(refs[:8], alts[:8], sw.STANDARD_NGS, sw.SOFTCLIP)
best = best_c = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    got = sw.align_batch(refs, alts, sw.STANDARD_NGS, sw.SOFTCLIP)
    best = min(best, time.perf_counter() - t0)
    best_c = min(best_c, sw.last_call_seconds)
sample = min(pairs, 200)
t0 = time.perf_counter()
want = [O.sw_align(r, a, O.SW_STANDARD_NGS, O.SW_SOFTCLIP)[1:] for r, a in zip(refs[:sample], alts[:sample])]
cpu = time.perf_counter() - t0
assert got[:sample] == want or os.environ.get("SW_BENCH_NOCHECK")
cpu_cells = sum(len(r) * len(a) for r, a in zip(refs[:sample], alts[:sample]))
print(f"{pairs} alignments, {cells:.3e} cells: GPU {1e3 * best_c:.2f} ms in gatk_sw_align incl. copies ({cells / best_c / 1e9:.1f} GCUPS), {1e3 * best:.2f} ms with the Python marshalling; "
      f"scalar CPU oracle {cpu_cells / cpu / 1e9:.3f} GCUPS on one thread ({sample} alignments)")
