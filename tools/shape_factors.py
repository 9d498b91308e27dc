"""Which property of HaplotypeCaller-shaped work costs the stream kernel its rate: resident batches (kernels only) that
go from the 4096 x 64, 151 / 600 batch towards that shape one property at a time.  usage: shape_factors.py [REPS]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hmm = HipLoglessPairHMM(0)
hmm.set_option("timing", 1)


def run(label, regions):
    cells = sum(r.cells for r in regions)
    res = hmm.upload(regions)
    for _ in range(2):
        res.compute(True)
    walls = []
    fp32 = []
    for _ in range(reps):
        t0 = time.perf_counter()
        res.compute(True)
        walls.append(time.perf_counter() - t0)
        fp32.append(hmm.stats()["fp32_kernel_ms"])
    st = hmm.stats()
    w = float(np.median(walls))
    k = float(np.median(fp32)) * 1e-3
    print(f"{label:58s} {cells:.3e} cells, pass {w * 1e3:7.3f} ms, forward kernels {k * 1e3:7.3f} ms -> {cells / k / 1e9:6.0f} GCUPS, K {st['main_rows_per_lane']}, "
          f"lane rows {st['lane_rows_used'] / st['lane_rows_allocated']:.3f}, items {st['wave_tasks']}", flush=True)
    res.free()


def many(n, nr, nh, rl, hl, seed=3, **kw):
    return [synth.make_batch(nr, nh, rl, hl, seed=seed + i, **kw) for i in range(n)]


run("4096 x 64, 151 / 600", [synth.make_batch(4096, 64, 151, 600, seed=1)])
run("4096 x 64, 151 / 375", [synth.make_batch(4096, 64, 151, 375, seed=1)])
run("64 regions of 64 x 64, 151 / 375", many(64, 64, 64, 151, 375))
run("256 regions of 64 x 16, 151 / 375", many(256, 64, 16, 151, 375))
run("455 regions of 64 x 9, 151 / 375", many(455, 64, 9, 151, 375))
run("455 regions of 64 x 9, 151 / 250-500", [synth.make_batch(64, 9, 151, 250 + (i * 37) % 251, seed=3 + i) for i in range(455)])
run("1820 regions of 64 x 9, 151 / 250-500", [synth.make_batch(64, 9, 151, 250 + (i * 37) % 251, seed=3 + i) for i in range(1820)])
run("1820 regions of 64 x 2-16, 151 / 250-500", [synth.make_batch(64, 2 + (i * 5) % 15, 151, 250 + (i * 37) % 251, seed=3 + i) for i in range(1820)])
run("1820 regions of 30-100 x 2-16, 151 / 250-500", [synth.make_batch(30 + (i * 7) % 71, 2 + (i * 5) % 15, 151, 250 + (i * 37) % 251, seed=3 + i) for i in range(1820)])
run("1820 regions of 30-100 x 2-16, 146-151 / 250-500", [synth.make_batch(30 + (i * 7) % 71, 2 + (i * 5) % 15, 151, 250 + (i * 37) % 251, seed=3 + i, len_jitter=5) for i in range(1820)])
run("2000 regions of synth.make_regions (121-151)", synth.make_regions(2000, seed=5))
hmm.close()
