"""One gatk_pairhmm_compute_regions call over N HaplotypeCaller-shaped regions (synth.make_regions, seed 5), repeated on a
warm handle: wall time per call, GCUPS, and the stats of the last call.  Options as name=value.  usage: hc_call.py [N] [REPS] [name=value ...]"""
import os
import sys
import time

import numpy as np

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
call = hmm.prepare_regions(regions)
for _ in range(3):
    call.compute()
walls = []
for _ in range(reps):
    t0 = time.perf_counter()
    call.compute()
    walls.append(time.perf_counter() - t0)
st = hmm.stats()
w = float(np.median(walls))
util = st["lane_rows_used"] / st["lane_rows_allocated"] if st["lane_rows_allocated"] else float("nan")
print(f"{' '.join(sys.argv[3:]) or 'defaults':32s} regions={n} cells={cells:.4e} median {w * 1e3:.3f} ms (best {min(walls) * 1e3:.3f}) -> {cells / w / 1e9:.0f} GCUPS "
      f"({cells * 12 / w / 157.3e12:.3f} of the fp32 roof); plan {st['host_plan_ms']:.2f} ms, upload {st['host_upload_ms']:.2f} ms, lane rows used {util:.3f}, "
      f"fp32 {st['fp32_kernel_ms']:.2f} ms fp64 {st['fp64_kernel_ms']:.2f} ms, rerun pairs {st['rerun_pairs']}")
hmm.close()
