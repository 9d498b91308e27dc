"""How many waves of the stream pass are resident over time: runs a shape on the profiling build
(make -C gatk_amd/csrc phase), reads its per-wave log and prints, per launch (rows per lane), the launch's length, the
mean number of resident waves and the resident waves at ten points of the launch.
usage: wave_timeline.py [c3 | hc | regions:N:READS:HAPS:HAPLEN] [name=value ...]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.environ.get("PAIRHMM_PHASE_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exp_build", "libphase.so")
os.environ["GATK_PAIRHMM_LIB"] = lib
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "hc"
hmm = HipLoglessPairHMM(0)
for kv in sys.argv[2:]:
    name, value = kv.split("=")
    hmm.set_option(name, float(value))
if shape == "c3":
    batches = [synth.config_c3()]
elif shape == "hc":
    batches = synth.make_regions(2000, seed=5)
else:
    _, n, nr, nh, hl = shape.split(":")
    batches = [synth.make_batch(int(nr), int(nh), 151, int(hl), seed=3 + i) for i in range(int(n))]
res = hmm.upload(batches)
res.compute()
dll = ctypes.CDLL(lib)
cap = 1 << 18
buf = np.zeros((cap, 4), dtype=np.uint32)
dll.gatk_pairhmm_debug_wave_log(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)), cap)
res.compute()
n = dll.gatk_pairhmm_debug_wave_log(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)), cap)
log = buf[:n].astype(np.int64)
if os.environ.get("PAIRHMM_WAVE_LOG_OUT"):
    np.save(os.environ["PAIRHMM_WAVE_LOG_OUT"], buf[:n])
t0 = log[:, 0].min()
start = (log[:, 0] - t0) & 0xFFFFFFFF
end = (log[:, 1] - t0) & 0xFFFFFFFF
rows = log[:, 2] & 0xFF
print(f"{shape}: {n} waves, {(end.max()) / 100:.0f} us from the first wave's start to the last wave's end")
for k in sorted(set(rows.tolist())):
    m = rows == k
    s, e = start[m], end[m]
    a, b = s.min(), e.max()
    length = b - a
    resident = (e - s).sum() / length
    pts = []
    for f in np.linspace(0.05, 0.95, 10):
        t = a + f * length
        pts.append(int(((s <= t) & (e > t)).sum()))
    dur = (e - s) / 100.0
    xcc = (log[m, 2] >> 8) & 0xF
    hw = log[m, 3]
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)  # CU id, shader engine id, XCD
    mid = (s < a + 0.6 * length) & (e > a + 0.2 * length)  # residency over the middle of the launch, per XCD
    per_xcd = []
    for x in sorted(set(xcc.tolist())):
        mx = xcc == x
        lo, hi = a + 0.2 * length, a + 0.6 * length
        per_xcd.append(int((np.minimum(e[mx], hi) - np.maximum(s[mx], lo)).clip(min=0).sum() / (hi - lo)))
    ncu = len(set(cu.tolist()))
    ends = [int((e[xcc == x].max() - a) / 100) for x in sorted(set(xcc.tolist()))]
    print(f"      resident waves per XCD over 20-60 % of the launch: {per_xcd}; distinct CUs seen {ncu}; last wave of each XCD ends at us {ends}")
    # the last tenth of the launch: how much of the slots' time is idle
    print(f"  {k:2d} rows per lane: {m.sum():6d} waves, launch {length / 100:8.1f} us, mean resident waves {resident:7.1f}; at 5..95 %: {pts}; "
          f"wave life us: median {np.median(dur):.1f}, longest {dur.max():.1f}")
hmm.close()
