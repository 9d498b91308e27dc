"""Quick GPU bring-up check: engine vs oracle on a few shapes, plus resident-path timing."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gatk_amd import synth  # noqa: E402
from gatk_amd.batch import pack  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402
from oracle import oracle as O  # noqa: E402


def cmp(name, hmm, b, ks=(0,)):
    rc, ref = O.compute_likelihoods(b, O.LOGLESS, fast_threads=8)
    assert rc == 0
    for k in ks:
        hmm.set_option("force_rows_per_lane", k)
        got = hmm.compute_batch(b)
        st = hmm.stats()
        fin = np.isfinite(ref)
        d = np.abs(got[fin] - ref[fin]).max() if fin.any() else 0.0
        same_inf = np.array_equal(np.isfinite(got), fin)
        print(f"{name:28s} K={k} reads={b.num_reads} haps={b.num_haps} max|d|={d:.3e} inf_match={same_inf} "
              f"tasks={st['wave_tasks']} rerun_tasks={st['rerun_tasks']} rerun_pairs={st['rerun_pairs']}", flush=True)
    hmm.set_option("force_rows_per_lane", 0)


def main():
    hmm = HipLoglessPairHMM(0)
    b = pack([("A", [30], [45], [45], [10])], ["A", "C", "AC"])
    print("KAT", hmm.compute_batch(b), "expect -0.0461920023 -3.5228787453 -0.3470771124", flush=True)
    cmp("tiny 3x2 r10 h20", hmm, synth.make_batch(3, 2, 10, 20, seed=1), ks=(0, 1, 2, 3))
    cmp("ragged", hmm, synth.make_batch(50, 7, 120, 300, seed=2, len_jitter=100), ks=(0, 2, 3, 5, 8))
    cmp("C1 128x16 101/400", hmm, synth.config_c1(), ks=(0, 2, 3, 4, 5, 6, 7, 8))
    cmp("C5-ish 64x8 250/1000", hmm, synth.config_c5(num_reads=64, num_haps=8), ks=(0, 4, 8))
    cmp("uniform quals", hmm, synth.make_batch(64, 8, 151, 500, seed=3, qual_model="uniform", gop_model="random"), ks=(0, 5))
    hmm.set_option("force_fp64", 1)
    cmp("C1 forced fp64", hmm, synth.config_c1(), ks=(0, 5))
    hmm.set_option("force_fp64", 0)
    hmm.set_option("emit_mode", 0)
    cmp("C1 compare-mode", hmm, synth.config_c1(), ks=(0, 5))
    hmm.set_option("emit_mode", -1)

    # resident timing on a C3 slice and full C3
    hmm.set_option("timing", 1)
    for nreads, nhaps in ((4096, 64),):
        t0 = time.time()
        b = synth.config_c3(num_reads=nreads, num_haps=nhaps)
        t1 = time.time()
        for k in (0, 6, 8, 10, 12, 16, 5, 7):
            hmm.set_option("emit_mode", 0 if k < 0 else -1)
            k = abs(k)
            hmm.set_option("force_rows_per_lane", k)
            res = hmm.upload([b])
            res.compute(sync=True)
            t2 = time.time()
            for _ in range(3):
                res.compute(sync=True)
            t3 = time.time()
            st = hmm.stats()
            ms = st["fp32_kernel_ms"] + st["fp64_kernel_ms"] + st["aux_kernel_ms"]
            print(f"C3 {nreads}x{nhaps} K={k}: cells={b.cells:.3e} fp32={st['fp32_kernel_ms']:.3f}ms fp64={st['fp64_kernel_ms']:.3f}ms "
                  f"-> {b.cells / ms / 1e6:.1f} GCUPS (events)  wall/iter={(t3 - t2) / 3 * 1e3:.2f}ms  tasks={st['wave_tasks']} "
                  f"rerun={st['rerun_tasks']}  gen={t1 - t0:.1f}s", flush=True)
            if nreads == 512 and k == 0:
                out = res.download()[0]
                rc, ref = O.compute_likelihoods(b, O.LOGLESS, fast_threads=8)
                print("   C3 slice parity max|d| =", np.abs(out - ref).max(), flush=True)
            res.free()
    # host-pointer path on the large batch: planning + H2D + kernels + D2H
    hmm.set_option("force_rows_per_lane", 0)
    hmm.set_option("emit_mode", -1)
    b = synth.config_c3()
    hmm.compute_batch(b)
    t0 = time.time()
    for _ in range(5):
        hmm.compute_batch(b)
    dt = (time.time() - t0) / 5
    print(f"C3 host-pointer path (plan + H2D + kernels + D2H): {dt * 1e3:.2f} ms/call -> {b.cells / dt / 1e9:.1f} GCUPS", flush=True)
    regions = synth.make_regions(2000, seed=5)
    cells = sum(r.cells for r in regions)
    hmm.compute_regions(regions[:10])
    t0 = time.time()
    outs = hmm.compute_regions(regions)
    dt = time.time() - t0
    st = hmm.stats()
    print(f"2000 HC-shaped regions in one pass: cells={cells:.3e} pairs={st['pairs']} wall={dt * 1e3:.1f} ms -> {cells / dt / 1e9:.1f} GCUPS; "
          f"kernels fp32={st['fp32_kernel_ms']:.2f} ms fp64={st['fp64_kernel_ms']:.2f} ms -> {cells / (st['fp32_kernel_ms'] + st['fp64_kernel_ms'] + st['aux_kernel_ms']) / 1e6:.1f} GCUPS; "
          f"tasks={st['wave_tasks']} rerun={st['rerun_tasks']}", flush=True)
    hmm.close()
    print("gpu_check done")


if __name__ == "__main__":
    main()
