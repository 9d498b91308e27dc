"""Where a wave's cycles go: runs the 4096 x 64 (151 / 600) batch, or another shape, on the profiling build
(make -C gatk_amd/csrc phase) and prints the per-phase share of wave cycles.
usage: phase_clocks.py [READS] [ROWS_PER_LANE] [c3 | hc | regions:N:READS:HAPS:HAPLEN]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exp_build", "libphase.so")
os.environ["GATK_PAIRHMM_LIB"] = lib
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hmm = HipLoglessPairHMM(0)
if k:
    hmm.set_option("force_rows_per_lane", k)
shape = sys.argv[3] if len(sys.argv) > 3 else "c3"
if shape == "c3":
    batches = [synth.config_c3(num_reads=reads)]
elif shape == "hc":
    batches = synth.make_regions(2000, seed=5)
else:
    _, n, nr, nh, hl = shape.split(":")
    batches = [synth.make_batch(int(nr), int(nh), 151, int(hl), seed=3 + i) for i in range(int(n))]
cells = sum(b.cells for b in batches)
res = hmm.upload(batches)
res.compute()
dll = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * 8)()
dll.gatk_pairhmm_debug_phase_clocks(out, 1)
res.compute()
dll.gatk_pairhmm_debug_phase_clocks(out, 1)
if out[7]:
    print(f"wave lifetime: {out[4]} shader ticks / {out[7]} ticks of the 100 MHz counter -> {100.0 * out[4] / out[7]:.0f} MHz")
names = ["decode+gap list", "row constants+table", "blocks with a lane at a gap", "steady blocks"]
tot = sum(out[:4])  # slot 4 is the wave lifetime, which the four phases add up to
print(f"{shape}: tasks {out[6]}  cycles/task {tot / max(out[6], 1):.0f}  wave cycles per cell {tot / cells:.4f}")
for n, v in zip(names, out[:4]):
    print(f"  {n:28s} {100.0 * v / tot:5.1f} %   {v / max(out[6], 1):9.0f} cycles/task")
