"""Driver for a device timeline (rocprofv3 --kernel-trace --memory-copy-trace) of compute_regions calls of N regions:
N HaplotypeCaller-shaped regions per call, REPS calls on a warm handle.  usage: call_trace.py N [REPS]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import pairhmm, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = synth.make_regions(n, seed=5)
lib = pairhmm.load_library()
keep = []
cbs = (pairhmm.CBatch * n)(*[pairhmm.to_cbatch(b, keep) for b in g])
outs = [np.empty((b.num_reads, b.num_haps)) for b in g]
arr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
h = pairhmm.HipLoglessPairHMM(0)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    assert lib.gatk_pairhmm_compute_regions(h._h, n, cbs, arr) == 0
    ts.append(time.perf_counter() - t0)
    time.sleep(0.0005)  # a visible gap between calls in the trace
print(f"{n} regions per call: median {np.median(ts[5:]) * 1e6:.0f} us")
h.close()
