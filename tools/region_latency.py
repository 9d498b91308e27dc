"""Latency of one gatk_pairhmm_compute_likelihoods call per active region (how the reference's likelihood
engine calls a PairHMM: once per region and sample), on HaplotypeCaller-shaped regions."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import pairhmm, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
options = [a.split("=") for a in sys.argv[2:]]  # engine options as name=value
regions = synth.make_regions(n, seed=5)
lib = pairhmm.load_library()
hmm = pairhmm.HipLoglessPairHMM(0)
for name, value in options:
    hmm.set_option(name, float(value))
keep = []
calls = []
for b in regions:
    out = np.empty((b.num_reads, b.num_haps))
    calls.append((pairhmm.to_cbatch(b, keep), out))
for cb, out in calls[:20]:
    assert lib.gatk_pairhmm_compute_likelihoods(hmm._h, C.byref(cb), out.ctypes.data) == 0
keys = ("host_plan_ms", "host_upload_ms", "aux_kernel_ms", "fp32_kernel_ms", "fp64_kernel_ms")
parts = {k: 0.0 for k in keys}
hmm.set_option("timing", 1)
for cb, out in calls:  # where a call's time goes, from the engine's own clocks and HIP events
    assert lib.gatk_pairhmm_compute_likelihoods(hmm._h, C.byref(cb), out.ctypes.data) == 0
    st = hmm.stats()
    for k in keys:
        parts[k] += st.get(k, 0.0)
print("mean per call, us: " + ", ".join(f"{k[:-3]} {v / n * 1e3:.1f}" for k, v in parts.items()))
hmm.set_option("timing", 0)
lat = []
t_all = time.perf_counter()
for cb, out in calls:
    t0 = time.perf_counter()
    assert lib.gatk_pairhmm_compute_likelihoods(hmm._h, C.byref(cb), out.ctypes.data) == 0
    lat.append(time.perf_counter() - t0)
t_all = time.perf_counter() - t_all
cells = sum(b.cells for b in regions)
lat = np.array(lat) * 1e6
print(f"{n} regions one call each: median {np.median(lat):.0f} us, p90 {np.percentile(lat, 90):.0f} us, max {lat.max():.0f} us per call; "
      f"{cells / t_all / 1e9:.0f} GCUPS overall ({np.mean([b.num_reads * b.num_haps for b in regions]):.0f} pairs per region on average)")
hmm.close()
