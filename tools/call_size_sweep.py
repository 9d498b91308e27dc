"""Time gatk_pairhmm_compute_regions on a warm handle for calls of 1 to 2000 HaplotypeCaller-shaped regions.

usage: call_size_sweep.py [name=value ...]   engine options for the handle, e.g. pipeline_min_pairs=131072
Prints, per call size: pairs, median wall time per call and per region, overall GCUPS including host and PCIe,
and (second pass, with the engine's event timing on) the forward-kernel and planning times."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import pairhmm, synth  # noqa: E402

options = [(a.split("=")[0], float(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]
regions = synth.make_regions(2000, seed=5)
lib = pairhmm.load_library()
print("options:", options or "defaults")
for n in [int(x) for x in os.environ.get("SWEEP_SIZES", "1,2,4,8,16,32,64,128,256,512,1000,2000").split(",")]:
    g = regions[:n]
    pairs = sum(b.num_reads * b.num_haps for b in g)
    cells = sum(int(np.diff(b.read_offsets).sum()) * int(np.diff(b.hap_offsets).sum()) for b in g)
    keep = []
    cbs = (pairhmm.CBatch * n)(*[pairhmm.to_cbatch(b, keep) for b in g])
    outs = [np.empty((b.num_reads, b.num_haps)) for b in g]
    arr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    h = pairhmm.HipLoglessPairHMM(0)
    for name, value in options:
        h.set_option(name, value)
    line = f"{n:5d} regions {pairs:8d} pairs:"
    for timing in (0, 1):
        h.set_option("timing", timing)
        for _ in range(4):
            assert lib.gatk_pairhmm_compute_regions(h._h, n, cbs, arr) == 0
        reps = 15 if n <= 512 else 6
        ts, fp32, plan = [], 0.0, 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            assert lib.gatk_pairhmm_compute_regions(h._h, n, cbs, arr) == 0
            ts.append(time.perf_counter() - t0)
            st = h.stats()
            fp32 += st["fp32_kernel_ms"] / reps
            plan += st["host_plan_ms"] / reps
        med = float(np.median(ts))
        if timing == 0:
            line += f" {med * 1e6:8.0f} us per call, {med * 1e6 / n:6.1f} us per region, {cells / med / 1e9:6.0f} GCUPS;"
        else:
            line += f" with event timing: forward kernels {fp32 * 1e3:7.0f} us, planning {plan * 1e3:6.0f} us"
    h.close()
    print(line, flush=True)
