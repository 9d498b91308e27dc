"""A named batch (c3: 4096 x 64, 151 / 600 bases; c5: the long-read / indel stress batch, 4096 x 32, 250 / 1000 bases) through one
gatk_pairhmm_compute_likelihoods call on host arrays, repeated on a warm handle: wall time per call and GCUPS.
Options as name=value.  usage: config_call.py c3|c5 [REPS] [name=value ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
sys.argv = sys.argv[:1] + sys.argv[2:]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
b = synth.config_c5(num_reads=4096, num_haps=32) if name == "c5" else synth.config_c3()
hmm = HipLoglessPairHMM(0)
for kv in sys.argv[2:]:
    name, value = kv.split("=")
    hmm.set_option(name, float(value))
call = hmm.prepare_regions([b])
for _ in range(3):
    call.compute()
walls = []
for _ in range(reps):
    t0 = time.perf_counter()
    call.compute()
    walls.append(time.perf_counter() - t0)
st = hmm.stats()
w = float(np.median(walls))
print(f"{name} {' '.join(sys.argv[2:]) or 'defaults':40s} median {w * 1e3:.3f} ms (best {min(walls) * 1e3:.3f}) -> {b.cells / w / 1e9:.0f} GCUPS "
      f"({b.cells * 12 / w / 157.3e12:.3f} of the fp32 roof); rerun pairs {st['rerun_pairs']}, plan {st['host_plan_ms']:.2f} ms")
hmm.close()
