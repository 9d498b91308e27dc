"""Host-pointer entry point on the large batch: wall time per gatk_pairhmm_compute_likelihoods call
(plan + one H2D + kernels + D2H), against the kernel time of the same call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

b = synth.config_c3()
hmm = HipLoglessPairHMM(0)
hmm.set_option("timing", 1)
for kv in sys.argv[1:]:  # engine options as name=value
    name, value = kv.split("=")
    hmm.set_option(name, float(value))
hmm.compute_batch(b)
hmm.compute_batch(b)
best = 1e9
for _ in range(8):
    t0 = time.perf_counter()
    hmm.compute_batch(b)
    best = min(best, time.perf_counter() - t0)
st = hmm.stats()
print(f"{' '.join(sys.argv[1:]) or 'defaults'}: C3 host-pointer call: {1e3 * best:.2f} ms wall ({b.cells / best / 1e9:.0f} GCUPS); plan {st['host_plan_ms']:.2f} ms, upload {st['host_upload_ms']:.2f} ms, "
      f"kernels {st['fp32_kernel_ms'] + st['fp64_kernel_ms'] + st['aux_kernel_ms']:.2f} ms")
hmm.close()
