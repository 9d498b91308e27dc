"""Resident kernel times of a named synthetic configuration: c1 | c3 | c5 [reads haps], engine options as name=value."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
nums = [int(a) for a in sys.argv[2:] if a.isdigit()]
opts = [a for a in sys.argv[2:] if "=" in a]
make = {"c1": synth.config_c1, "c3": synth.config_c3, "c5": synth.config_c5}[name]
b = make(num_reads=nums[0], num_haps=nums[1]) if len(nums) == 2 else make()
hmm = HipLoglessPairHMM(0)
hmm.set_option("timing", 1)
for kv in opts:
    k, v = kv.split("=")
    hmm.set_option(k, float(v))
res = hmm.upload([b])
for _ in range(4):
    res.compute(sync=True)
st = hmm.stats()
tot = st["fp32_kernel_ms"] + st["fp64_kernel_ms"] + st["aux_kernel_ms"]
print(f"{name} {b.num_reads}x{b.num_haps} {' '.join(opts)}: cells={b.cells:.3e} fp32={st['fp32_kernel_ms']:.3f} ms fp64={st['fp64_kernel_ms']:.3f} ms aux={st['aux_kernel_ms']:.3f} ms "
      f"-> {b.cells / tot / 1e6:.0f} GCUPS; rerun items {st['rerun_tasks']}/{st['wave_tasks']} pairs {st['rerun_pairs']}/{st['pairs']}")
res.free()
hmm.close()
