#!/usr/bin/env python3
"""PairHMM GCUPS benchmark.

A step is one pass of the hot path over one synthetic batch that is already resident in HBM:
input validation, expansion of reads and haplotypes, the fp32 forward kernel, the conversion of its
sums to log10 and the fp64 rerun of underflowing work items, all enqueued through the C-ABI
(gatk_pairhmm_compute_resident).  The K timed steps are enqueued back to back on the engine's stream
and waited for once; the engine brackets every pass with HIP events on that stream, so the kernel
time in the roofline object is the average over exactly the timed passes.  The default workload is the
large-batch configuration the GCUPS metric is quoted on: 4096 reads x 64 haplotypes, 151 bp /
600 bp.  With N GPUs every rank scores its own batch of that shape (pairs are independent:
no data-path collective), so scaling is weak and value = N * cells / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CELL = 12          # 8 mul + 4 add of the linear-space three-state recurrence
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X: 256 CU x 4 SIMD x 32 lanes x 2 FLOP x 2.4 GHz
# HBM bytes of one fp32 forward launch on the default workload, from PMC counters collected in a run
# of their own (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_stream_kernel.txt); not measured live
PMC_TRAFFIC_BYTES_DEFAULT_WORKLOAD = 2 * 10287.9e3 + 6858.8e3


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def timed_sample(fn, batch, threads, target_seconds):
    """Run fn on the first reads of the batch: a probe to size the sample, then about target_seconds of work."""
    probe = batch.read_slice(0, min(batch.num_reads, 4 * threads))
    t0 = time.perf_counter()
    fn(probe)
    dt = max(time.perf_counter() - t0, 1e-6)
    rate = probe.cells / dt
    n = int(min(batch.num_reads, max(probe.num_reads, target_seconds * rate / (batch.cells / batch.num_reads))))
    n = max(threads, n - n % threads)
    sample = batch.read_slice(0, n)
    # the whole batch may take well under the target: repeat it
    reps = max(1, min(200, int(target_seconds * rate / sample.cells)))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(sample)
    dt = (time.perf_counter() - t0) / reps
    return sample, n, dt, reps


def cpu_baseline(batch, target_seconds=8.0):
    """CPU baselines on this host's cores, on a bounded sample of the same workload: the vectorised
    fp32-with-fp64-redo pass (oracle/pairhmm_simd.cpp, the shape of the AVX library GATK 3.x loads)
    is the reported value; the scalar fp64 restatement (the parity oracle) is timed beside it."""
    from oracle import oracle as O

    threads = max(1, min(16, os.cpu_count() or 1))

    def simd(b):
        assert O.simd_compute_likelihoods(b, threads)[0] == 0

    def scalar(b):
        assert O.compute_likelihoods(b, O.LOGLESS, fast_threads=threads)[0] == 0

    s1, n1, dt1, reps1 = timed_sample(simd, batch, threads, target_seconds)
    s2, n2, dt2, reps2 = timed_sample(scalar, batch, threads, target_seconds)
    return {"value": round(s1.cells / dt1 / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "port",
            "sample": f"first {n1} of {batch.num_reads} reads x {batch.num_haps} haplotypes of the same batch ({s1.cells:.3e} cells, "
                      f"{reps1} x {dt1:.2f} s), fp32 with fp64 redo, {O.simd_isa()} vectors over haplotypes, OpenMP over reads, on {cpu_model()}",
            "scalar_fp64_oracle": {"value": round(s2.cells / dt2 / 1e9, 3), "unit": "GCUPS", "cores": threads,
                                   "sample": f"first {n2} reads ({s2.cells:.3e} cells, {reps2} x {dt2:.2f} s), OpenMP over reads"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=4096)
    ap.add_argument("--haps", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the PairHMM engine has no CPU path")
    # Rehearsal of the multi-rank path on a box with fewer GPUs than ranks: GATK_BENCH_BACKEND=gloo keeps the
    # two reductions on the host and lets ranks share a device.  The driver's runs use RCCL, one rank per GPU.
    backend = os.environ.get("GATK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    reduce_device = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from gatk_amd import synth
    from gatk_amd.pairhmm import HipLoglessPairHMM

    # every rank scores a batch of the same shape (different seed): weak scaling
    batch = synth.config_c3(seed=synth.DEFAULT_SEED + rank, num_reads=args.reads, num_haps=args.haps)
    hmm = HipLoglessPairHMM(local_rank)
    hmm.set_option("timing", 1)
    res = hmm.upload([batch])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res.compute(sync=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res.compute(sync=False)
    res.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    st = hmm.stats()
    assert st["timed_passes"] == args.steps
    fp32_ms, fp64_ms, aux_ms = st["fp32_kernel_ms"], st["fp64_kernel_ms"], st["aux_kernel_ms"]
    rerun_pairs = st["rerun_pairs"] * args.steps  # every pass scores the same batch
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([float(batch.cells)], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = float(c.item())
    else:
        total_cells = float(batch.cells)

    # sanity: results of the last step are finite log10 likelihoods <= 0
    out = res.download()[0]
    assert out.shape == (batch.num_reads, batch.num_haps) and bool((out <= 0).all())

    if rank == 0:
        steps = max(args.steps, 1)
        kern_s = fp32_ms / steps * 1e-3
        achieved = batch.cells * FLOP_PER_CELL / kern_s / 1e12
        line = {
            "metric": "PairHMM GCUPS", "value": round(total_cells * steps / elapsed / 1e9, 2), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"large-batch PairHMM: {args.reads} reads x {args.haps} haplotypes, 151 bp / 600 bp, per GPU",
                       "cells_per_gpu_step": batch.cells, "pairs_per_gpu_step": batch.num_reads * batch.num_haps,
                       "parallelism": f"independent batches on {world} GPU(s), host gather, no collective"},
            "roofline": {"bound": "valu", "bound_detail": "fp32 vector ALU (a recurrence: neither HBM- nor MFMA-bound)",
                         "kernel": "fwd_stream<10, 2> (fp32 forward pass)", "achieved": round(achieved, 2), "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
                         "traffic": PMC_TRAFFIC_BYTES_DEFAULT_WORKLOAD if (args.reads, args.haps) == (4096, 64) else None,
                         "flop_per_cell": FLOP_PER_CELL, "kernel_ms": round(fp32_ms / steps, 4),
                         "kernel_gcups": round(batch.cells / kern_s / 1e9, 1)},
            "other_kernels_ms": round(aux_ms / steps, 4),
            "fp64_rerun": {"pairs_per_step": rerun_pairs // steps, "rate": round(rerun_pairs / steps / (batch.num_reads * batch.num_haps), 6),
                           "kernel_ms": round(fp64_ms / steps, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batch)
        print(json.dumps(line), flush=True)
    res.free()
    hmm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
