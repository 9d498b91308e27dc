"""Where a wave's cycles go in a region-sized call (one-haplotype fp32 pass): runs HaplotypeCaller-shaped regions one
call each on the profiling build (make -C gatk_amd/csrc phase) and prints the per-phase share of wave cycles.  A wave
keeps its shares in registers and adds them to the build's counters when its work item is over (atomics in the middle
of an item stood in front of the wave's later global loads and tripled the set-up's share)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.environ.get("PHASE_LIB", os.path.join(ROOT, "exp_build", "libphase.so"))
os.environ["GATK_PAIRHMM_LIB"] = lib
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
regions = synth.make_regions(n, seed=5)
hmm = HipLoglessPairHMM(0)
dll = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * 8)()
for b in regions[:10]:
    hmm.compute_batch(b)
dll.gatk_pairhmm_debug_phase_clocks(out, 1)
for b in regions:
    hmm.compute_batch(b)
dll.gatk_pairhmm_debug_phase_clocks(out, 1)
names = ["decode + haplotype image", "row constants + table", "pipeline fill", "steady blocks", "drain", "termination"]
tot = sum(out[:6])
print(f"{n} region-sized calls: {out[6]} work items, {tot / max(out[6], 1):.0f} cycles per item")
for name, v in zip(names, out[:6]):
    print(f"  {name:28s} {100.0 * v / tot:5.1f} %   {v / max(out[6], 1):9.0f} cycles per item")
hmm.close()
