"""Kernel time of a HaplotypeCaller-shaped call (synth.make_regions) under engine options given as name=value."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
jitter = [int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("jitter=")]  # reads of length 151 - jitter .. 151
sys.argv = sys.argv[:2] + [a for a in sys.argv[2:] if not a.startswith("jitter=")]
regions = synth.make_regions(n, seed=5, read_len_jitter=jitter[0] if jitter else 30)
cells = sum(r.cells for r in regions)
hmm = HipLoglessPairHMM(0)
hmm.set_option("timing", 1)
for kv in sys.argv[2:]:
    name, value = kv.split("=")
    hmm.set_option(name, float(value))
res = hmm.upload(regions)
for _ in range(4):
    res.compute(sync=True)
st = hmm.stats()
ms = st["fp32_kernel_ms"]
print(f"{' '.join(sys.argv[2:]) or 'defaults':40s} regions={n} cells={cells:.3e} fp32={ms:.2f} ms aux={st['aux_kernel_ms']:.2f} fp64={st['fp64_kernel_ms']:.2f} "
      f"-> {cells / ms / 1e6:.0f} GCUPS  items={st['wave_tasks']} rerun={st['rerun_tasks']}")
res.free()
hmm.close()
