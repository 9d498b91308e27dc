"""Where a wave's cycles go: runs the 4096 x 64 (151 / 600) batch on the profiling build
(make -C gatk_amd/csrc phase) and prints the per-phase share of wave cycles."""
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
b = synth.config_c3(num_reads=reads)
res = hmm.upload([b])
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
print(f"tasks {out[6]}  cycles/task {tot / max(out[6], 1):.0f}  kernel_ms {hmm.stats()['fp32_kernel_ms']:.3f}")
for n, v in zip(names, out[:4]):
    print(f"  {n:28s} {100.0 * v / tot:5.1f} %   {v / max(out[6], 1):9.0f} cycles/task")
