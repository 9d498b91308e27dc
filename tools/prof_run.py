"""Small driver for profilers: a few resident passes of the large-batch workload, no torch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
b = synth.config_c3(num_reads=reads)
hmm = HipLoglessPairHMM(0)
hmm.set_option("timing", 1)
hmm.set_option("force_rows_per_lane", k)
for kv in sys.argv[4:]:  # further engine options as name=value
    name, value = kv.split("=")
    hmm.set_option(name, float(value))
res = hmm.upload([b])
for _ in range(iters):
    res.compute(sync=True)
st = hmm.stats()
print(f"cells={b.cells} fp32_ms={st['fp32_kernel_ms']:.3f} gcups={b.cells / st['fp32_kernel_ms'] / 1e6:.1f} tasks={st['wave_tasks']}")
res.free()
hmm.close()
