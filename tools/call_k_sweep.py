"""Forced rows-per-lane sweep on calls of 4 to 128 HaplotypeCaller-shaped regions (one compute_regions call each on a warm
handle): median wall time per call for the planner's own choice (K=0) and for forced values; in brackets the rows per lane
of the call's main launch and whether it was the stream pass (s) or the one-haplotype pass (d)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import pairhmm, synth
options = [(a.split("=")[0], float(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]  # engine options for every handle
regions = synth.make_regions(256, seed=5)
print("options:", options or "defaults")
lib = pairhmm.load_library()
for n in (4, 8, 12, 16, 24, 32, 48, 64, 128):
    g = regions[:n]
    keep = []
    cbs = (pairhmm.CBatch * n)(*[pairhmm.to_cbatch(b, keep) for b in g])
    outs = [np.empty((b.num_reads, b.num_haps)) for b in g]
    arr = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    row = []
    for k in (0, 3, 4, 5, 6, 8, 10):
        h = pairhmm.HipLoglessPairHMM(0)
        h.set_option("force_rows_per_lane", k)
        for name, value in options:
            h.set_option(name, value)
        for _ in range(10):
            assert lib.gatk_pairhmm_compute_regions(h._h, n, cbs, arr) == 0
        ts = []
        for _ in range(60):
            t0 = time.perf_counter(); lib.gatk_pairhmm_compute_regions(h._h, n, cbs, arr); ts.append(time.perf_counter() - t0)
        st = h.stats()
        row.append(f"K={k}:{np.median(ts)*1e6:6.0f}us(k{st['main_rows_per_lane']}{'s' if st['main_stream_pass'] else 'd'})")
        h.close()
    print(f"{n:4d} regions: " + "  ".join(row), flush=True)
