"""Where does the host time of a many-region call go? (Python marshalling vs C planning/upload vs kernels)"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gatk_amd import pairhmm, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t0 = time.time()
regions = synth.make_regions(n, seed=5)
print(f"generate {n} regions: {time.time() - t0:.2f} s; reads={sum(r.num_reads for r in regions)} pairs={sum(r.num_reads * r.num_haps for r in regions)}")
hmm = pairhmm.HipLoglessPairHMM(0)
hmm.set_option("timing", 1)
hmm.compute_regions(regions[:8])
lib = pairhmm.load_library()
for rep in range(3):
    t0 = time.time()
    keep = []
    cbs = (pairhmm.CBatch * len(regions))(*[pairhmm.to_cbatch(b, keep) for b in regions])
    outs = [np.empty((b.num_reads, b.num_haps)) for b in regions]
    arr = (C.c_void_p * len(regions))(*[o.ctypes.data for o in outs])
    t1 = time.time()
    r = C.c_void_p()
    rc = lib.gatk_pairhmm_upload(hmm._h, len(regions), cbs, C.byref(r))
    assert rc == 0
    t2 = time.time()
    rc = lib.gatk_pairhmm_compute_resident(hmm._h, r, 1)
    assert rc == 0
    t3 = time.time()
    rc = lib.gatk_pairhmm_download(hmm._h, r, arr)
    assert rc == 0
    t4 = time.time()
    lib.gatk_pairhmm_resident_free(hmm._h, r)
    st = hmm.stats()
    cells = sum(b.cells for b in regions)
    print(f"marshal {1e3 * (t1 - t0):.1f} ms | plan+upload {1e3 * (t2 - t1):.1f} ms | compute {1e3 * (t3 - t2):.1f} ms "
          f"(kernels {st['fp32_kernel_ms'] + st['fp64_kernel_ms'] + st['aux_kernel_ms']:.1f} ms, {cells / (st['fp32_kernel_ms'] + st['fp64_kernel_ms'] + st['aux_kernel_ms']) / 1e6:.0f} GCUPS) | download {1e3 * (t4 - t3):.1f} ms | tasks {st['wave_tasks']}")
hmm.close()
