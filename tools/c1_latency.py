"""Latency of one computeLikelihoods call on the 128 x 16 batch (host arrays in and out) under a few engine options:
where the call's time goes (the engine's own clocks and HIP events) and what the rows-per-lane choice costs."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth, pairhmm
b = synth.config_c1()
lib = pairhmm.load_library()
for opts in ([], [("signal_download_bytes", 0)], [("force_rows_per_lane", 2)], [("force_rows_per_lane", 3)], [("force_rows_per_lane", 4)], [("force_rows_per_lane", 6)], [("stream_mode", 1)]):
    hmm = pairhmm.HipLoglessPairHMM(0)
    for k, v in opts:
        hmm.set_option(k, float(v))
    keep = []
    cb = pairhmm.to_cbatch(b, keep)
    out = np.empty((b.num_reads, b.num_haps))
    for _ in range(20):
        assert lib.gatk_pairhmm_compute_likelihoods(hmm._h, C.byref(cb), out.ctypes.data) == 0
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        lib.gatk_pairhmm_compute_likelihoods(hmm._h, C.byref(cb), out.ctypes.data)
        ts.append(time.perf_counter() - t0)
    hmm.set_option("timing", 1)
    lib.gatk_pairhmm_compute_likelihoods(hmm._h, C.byref(cb), out.ctypes.data)
    st = hmm.stats()
    print(opts, f"median {np.median(ts)*1e6:.1f} us min {np.min(ts)*1e6:.1f}; K={st['main_rows_per_lane']} stream={st['main_stream_pass']} fp32 {st['fp32_kernel_ms']*1e3:.1f} us aux {st['aux_kernel_ms']*1e3:.1f} plan {st['host_plan_ms']*1e3:.1f} upload {st['host_upload_ms']*1e3:.1f}")
    hmm.close()
