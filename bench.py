#!/usr/bin/env python3
"""PairHMM GCUPS benchmark.

A step is one pass of the hot path over one synthetic batch: input validation, expansion of reads and
haplotypes, the fp32 forward kernel, the conversion of its sums to log10 and the fp64 rerun of underflowing
pairs, all through the C-ABI.  The default workload is the large-batch configuration the GCUPS metric is quoted
on: 4096 reads x 64 haplotypes, 151 bp / 600 bp.

Two timed legs of K steps each, both bracketed by barrier + device synchronisation:
  * resident  (`value`): the batch is already in HBM when the clock starts (gatk_pairhmm_compute_resident);
    the K passes are enqueued back to back and waited for once.  The engine brackets every pass with HIP events
    on its own stream, so the kernel time of the roofline object is the average over exactly these passes.
  * end to end (`end_to_end`): K calls of gatk_pairhmm_compute_likelihoods on host arrays: checks, planning, the
    host-to-device copy, every kernel, the copy back into the caller's array.  Includes PCIe.

On rank 0 of a one-GPU run the other workloads follow, each through the host-pointer entry points (`configs`):
the 128 x 16 CPU-sized batch, the 250 / 1000 high-indel batch with its fp64 rerun rate, a HaplotypeCaller-shaped
stream of 2000 active regions in one compute_regions call, one region per call (blocking, and with two and four regions
in flight through submit / fetch), a region of eight samples through the region-level hook, and region-sized calls
from 1, 4 and 16 host threads; every entry carries its fraction of the fp32 vector roof.  Then the CPU baseline on a
bounded sample.  With N GPUs every rank scores its own batch of the default shape (pairs are independent: no
data-path collective), so scaling is weak and value = N * cells / max-over-ranks time; the `c4` object adds the
whole-genome-shaped 10 000 regions, sharded over the ranks with the host gather inside the clock (strong scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CELL = 12          # 8 mul + 4 add of the linear-space three-state recurrence
EXECUTED_FLOP_PER_CELL = 9  # what the rescaled fp32 cell executes: 4 fused multiply-adds and 1 multiply
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X: 256 CU x 4 SIMD x 32 lanes x 2 FLOP x 2.4 GHz
PEAK_FP64_VALU_TFLOPS = 78.6   # the fp64 passes: half the fp32 vector rate
# HBM bytes of one fp32 forward launch on the default workload, from PMC counters collected in a run
# of their own (2 x FETCH_SIZE + WRITE_SIZE); not measured live
# (FETCH_SIZE counts a wide coalesced read at half its bytes on gfx950: doubled; both counters are in KiB)
PMC_TRAFFIC_BYTES_DEFAULT_WORKLOAD = (2 * 10285.2 + 6588.8) * 1024
PMC_TRAFFIC_SOURCE = "profiles/r03_pmc_stream_kernel.txt (separate rocprofv3 --pmc passes over tools/prof_run.py 4096 4; not measured in this run)"
EMIT_MODES = {0: "compare", 1: "compare+N", 2: "table", 3: "table+N"}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def host_threads():
    """Threads the CPU legs may use: every core this process is allowed (scheduler affinity), capped by the
    OMP_NUM_THREADS the GPU box sets for its one-GPU share of the machine."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    omp = os.environ.get("OMP_NUM_THREADS", "")
    limit = int(omp) if omp.isdigit() and int(omp) > 0 else allowed
    return max(1, min(allowed, limit)), allowed, omp or None


def timed_sample(fn, batch, threads, target_seconds):
    """Run fn on the first reads of the batch: a probe to size the sample, then about target_seconds of work."""
    probe = batch.read_slice(0, min(batch.num_reads, 4 * threads))
    t0 = time.perf_counter()
    fn(probe)
    dt = max(time.perf_counter() - t0, 1e-6)
    rate = probe.cells / dt
    n = int(min(batch.num_reads, max(probe.num_reads, target_seconds * rate / (batch.cells / batch.num_reads))))
    n = max(threads, n - n % threads)
    n = min(n, batch.num_reads)
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

    threads, allowed, omp = host_threads()

    def simd(b):
        assert O.simd_compute_likelihoods(b, threads)[0] == 0

    def scalar(b):
        assert O.compute_likelihoods(b, O.LOGLESS, fast_threads=threads)[0] == 0

    def scalar_one_thread(b):
        assert O.compute_likelihoods(b, O.LOGLESS, fast_threads=1)[0] == 0

    s1, n1, dt1, reps1 = timed_sample(simd, batch, threads, target_seconds)
    s2, n2, dt2, reps2 = timed_sample(scalar, batch, threads, target_seconds)
    s3, n3, dt3, reps3 = timed_sample(scalar_one_thread, batch, 1, target_seconds / 2)
    return {"value": round(s1.cells / dt1 / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "port",
            "host": {"cpu": cpu_model(), "os_cpu_count": os.cpu_count(), "cpus_in_affinity_mask": allowed, "OMP_NUM_THREADS": omp,
                     "threads_used": threads,
                     "note": "every core this process may use; the GPU box gives a one-GPU job OMP_NUM_THREADS cores of the machine"},
            "sample": f"first {n1} of {batch.num_reads} reads x {batch.num_haps} haplotypes of the same batch ({s1.cells:.3e} cells, "
                      f"{reps1} x {dt1:.2f} s), fp32 with fp64 redo, {O.simd_isa()} vectors over haplotypes, OpenMP over reads, on {cpu_model()}",
            "scalar_fp64_oracle": {"value": round(s2.cells / dt2 / 1e9, 3), "unit": "GCUPS", "cores": threads,
                                   "sample": f"first {n2} reads ({s2.cells:.3e} cells, {reps2} x {dt2:.2f} s), OpenMP over reads"},
            "scalar_fp64_oracle_one_thread": {"value": round(s3.cells / dt3 / 1e9, 3), "unit": "GCUPS", "cores": 1,
                                              "sample": f"first {n3} reads ({s3.cells:.3e} cells, {reps3} x {dt3:.2f} s)"}}


class HostCall:
    """One prepared C-ABI call on host arrays: ctypes views and the output arrays are made once, the clock covers
    the library call only."""

    def __init__(self, hmm, regions, single):
        from gatk_amd import pairhmm
        self.lib, self.hmm, self.regions, self.single = pairhmm.load_library(), hmm, regions, single
        self.keep = []
        self.outs = [np.full((b.num_reads, b.num_haps), np.nan, dtype=np.float64) for b in regions]
        self.cbs = (pairhmm.CBatch * len(regions))(*[pairhmm.to_cbatch(b, self.keep) for b in regions])
        self.arr = (C.c_void_p * len(regions))(*[o.ctypes.data if o.size else None for o in self.outs])
        self.cells = sum(b.cells for b in regions)
        self.pairs = sum(b.num_reads * b.num_haps for b in regions)

    def __call__(self):
        if self.single:
            rc = self.lib.gatk_pairhmm_compute_likelihoods(self.hmm._h, C.byref(self.cbs[0]), self.arr[0])
        else:
            rc = self.lib.gatk_pairhmm_compute_regions(self.hmm._h, len(self.regions), self.cbs, self.arr)
        if rc != 0:
            raise RuntimeError(self.lib.gatk_pairhmm_last_error().decode())

    def check(self):
        for o in self.outs:
            assert not np.isnan(o).any() and bool((o <= 0).all())


def time_calls(call, reps, warm=3):
    for _ in range(warm):
        call()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    call.check()
    return float(np.median(ts)), float(np.min(ts))


def roof_frac(cells, seconds, peak_tflops=PEAK_FP32_VALU_TFLOPS):
    """Fraction of the vector-ALU roof: cells x 12 FLOP over the wall (or kernel) time, against the peak."""
    return round(cells * FLOP_PER_CELL / seconds / (peak_tflops * 1e12), 4)


def leg(call, reps, what):
    med, best = time_calls(call, reps)
    st = call.hmm.stats()
    out = {"workload": what, "pairs": call.pairs, "cells": call.cells, "wall_ms": round(med * 1e3, 4), "best_ms": round(best * 1e3, 4),
           "gcups": round(call.cells / med / 1e9, 1), "roofline_frac": roof_frac(call.cells, med), "calls_timed": reps,
           "path": "host arrays in, host arrays out (plan + H2D + kernels + D2H)"}
    if st["lane_rows_allocated"]:
        out["lane_rows_used"] = round(st["lane_rows_used"] / st["lane_rows_allocated"], 4)
    return out


def kernel_split(device, batch):
    """fp32 and fp64 kernel times of one resident pass over `batch` (HIP events on the engine's stream), each as a
    fraction of its own vector-ALU peak: the fp32 passes over every cell, the fp64 passes over the cells of the pairs
    they redo."""
    from gatk_amd.pairhmm import HipLoglessPairHMM
    hmm = HipLoglessPairHMM(device)
    try:
        hmm.set_option("timing", 1)
        res = hmm.upload([batch])
        for _ in range(3):
            res.compute(sync=True)
        st = hmm.stats()
        res.free()
    finally:
        hmm.close()
    redo_cells = st["rerun_pairs"] * (batch.cells / max(batch.num_reads * batch.num_haps, 1))
    out = {"fp32_kernel_ms": round(st["fp32_kernel_ms"], 4), "fp32_roofline_frac": roof_frac(batch.cells, st["fp32_kernel_ms"] * 1e-3),
           "fp64_kernel_ms": round(st["fp64_kernel_ms"], 4), "fp64_pairs": st["rerun_pairs"], "other_kernels_ms": round(st["aux_kernel_ms"], 4)}
    if st["rerun_pairs"] and st["fp64_kernel_ms"] > 0:
        out["fp64_roofline_frac"] = roof_frac(redo_cells, st["fp64_kernel_ms"] * 1e-3, PEAK_FP64_VALU_TFLOPS)
        out["fp64_peak_tflops"] = PEAK_FP64_VALU_TFLOPS
    return out


def in_flight_leg(hmm, regions, depth):
    """One caller thread keeps `depth` regions on the device: gatk_pairhmm_submit for region n + depth - 1 goes out
    before gatk_pairhmm_fetch of region n."""
    from gatk_amd import pairhmm
    lib = pairhmm.load_library()
    calls = [HostCall(hmm, [b], True) for b in regions]
    tickets = [C.c_int64() for _ in calls]

    def submit(k):
        if lib.gatk_pairhmm_submit(hmm._h, 1, calls[k].cbs, calls[k].arr, C.byref(tickets[k])) != 0:
            raise RuntimeError(lib.gatk_pairhmm_last_error().decode())

    def fetch(k):
        if lib.gatk_pairhmm_fetch(hmm._h, tickets[k]) != 0:
            raise RuntimeError(lib.gatk_pairhmm_last_error().decode())

    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(len(calls)):
            if k >= depth:
                fetch(k - depth)
            submit(k)
        for k in range(max(0, len(calls) - depth), len(calls)):
            fetch(k)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    for c in calls:
        c.check()
    cells = sum(c.cells for c in calls)
    return {"workload": f"{len(calls)} of those regions from one caller thread, gatk_pairhmm_submit / gatk_pairhmm_fetch with {depth} regions in flight",
            "us_per_region": round(best * 1e6 / len(calls), 1), "gcups": round(cells / best / 1e9, 1), "roofline_frac": roof_frac(cells, best)}


def region_hook_leg(hmm, region, samples):
    """A region of `samples` samples through gatk_pairhmm_initialize_region + one computeLikelihoods call per sample +
    finalizeRegion (PairHMM.java:121-139,190), against the same reads as one sample."""
    from gatk_amd import pairhmm
    from gatk_amd.batch import Batch
    lib = pairhmm.load_library()
    out = {}
    for s in (1, samples):
        cuts = np.linspace(0, region.num_reads, s + 1).astype(int)
        parts = [region.read_slice(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:])]
        ahead = [Batch(p.read_bases, p.read_quals, p.ins_gop, p.del_gop, None, p.read_offsets, p.hap_bases, p.hap_offsets) for p in parts]
        keep = []
        cbs_ahead = (pairhmm.CBatch * s)(*[pairhmm.to_cbatch(b, keep) for b in ahead])
        calls = [HostCall(hmm, [p], True) for p in parts]
        best = None
        for rep in range(40):
            t0 = time.perf_counter()
            if lib.gatk_pairhmm_initialize_region(hmm._h, s, cbs_ahead, 1000, 16000) != 0:
                raise RuntimeError(lib.gatk_pairhmm_last_error().decode())
            for c in calls:
                c()
            lib.gatk_pairhmm_finalize_region(hmm._h)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        for c in calls:
            c.check()
        out[s] = best
    return {"workload": f"one region of {region.num_reads} reads x {region.num_haps} haplotypes as {samples} samples: initialize_region, "
                        f"{samples} computeLikelihoods calls, finalizeRegion", "us_per_region": round(out[samples] * 1e6, 1),
            "same_reads_as_one_sample_us": round(out[1] * 1e6, 1), "ratio": round(out[samples] / out[1], 3),
            "region_hits": hmm.stats()["region_hits"]}


def shared_device_legs():
    """Region-sized calls from several host threads, each with its own handle, on one GPU (tools/thread_bench.cpp, a
    C++ driver: Python threads would measure the interpreter lock).  Skipped when the driver has not been built."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "thread_bench_bin")
    if not os.path.exists(exe):
        return {"skipped": "tools/thread_bench_bin is not built (g++ -O2 -pthread tools/thread_bench.cpp -Lgatk_amd -lgatk_pairhmm_hip)"}
    out = {}
    modes = {"blocking_calls": 0, "blocking_calls_sharing_the_dispatcher": 1, "two_in_flight_per_thread": -2, "three_in_flight_per_thread": -3}
    for threads, names in ((1, ("blocking_calls", "two_in_flight_per_thread")), (4, tuple(modes)), (16, ("blocking_calls_sharing_the_dispatcher", "two_in_flight_per_thread"))):
        for name in names:
            try:
                txt = subprocess.run([exe, "2000", str(threads), str(modes[name])], capture_output=True, text=True, timeout=120, check=True).stdout
                us = float(txt.split(" us per region")[0].split()[-1])
                gcups = float(txt.split(" GCUPS")[0].split()[-1])
                out[f"{threads}_threads_{name}"] = {"us_per_region": us, "gcups": gcups, "roofline_frac": round(gcups * 1e9 * FLOP_PER_CELL / (PEAK_FP32_VALU_TFLOPS * 1e12), 4)}
            except Exception as e:  # never lose the line over the extra leg
                out[f"{threads}_threads_{name}"] = {"error": repr(e)}
    out["workload"] = "2000 synthetic HaplotypeCaller-sized regions (30-100 reads of 121-151 bases, 2-16 haplotypes of 250-500 bases), one region per call"
    return out


def other_configs(device, hc_regions):
    """The other named workloads on one GPU, each through the host-pointer entry points on a default handle."""
    from gatk_amd import synth
    from gatk_amd.pairhmm import HipLoglessPairHMM

    out = {}
    hmm = HipLoglessPairHMM(device)
    try:
        out["c1"] = leg(HostCall(hmm, [synth.config_c1()], True), 50, "128 reads (101 bp) x 16 haplotypes (~400 bp), one computeLikelihoods call")
        c5 = synth.config_c5(num_reads=4096, num_haps=32)
        call = HostCall(hmm, [c5], True)
        out["c5"] = leg(call, 10, "long-read / indel stress: 4096 reads (250 bp, 4 % indels, mean Q20) x 32 haplotypes (1000 bp)")
        st = hmm.stats()
        out["c5"]["fp64_rerun"] = {"pairs": st["rerun_pairs"], "rate": round(st["rerun_pairs"] / max(call.pairs, 1), 6)}
        out["c5"]["kernels"] = kernel_split(device, c5)
        try:  # reads with more rows than a wave holds: the fp64 long-read pass (no HaplotypeCaller input; reported for completeness)
            long_reads = synth.make_batch(256, 16, 2000, 2200, seed=11)
            call = HostCall(hmm, [long_reads], True)
            out["long_reads"] = leg(call, 3, "256 reads (2000 bp) x 16 haplotypes (2200 bp): every pair through the fp64 long-read pass")
            out["long_reads"]["roofline_frac"] = roof_frac(call.cells, out["long_reads"]["wall_ms"] * 1e-3, PEAK_FP64_VALU_TFLOPS)
            out["long_reads"]["roofline_peak_tflops"] = PEAK_FP64_VALU_TFLOPS
        except Exception as e:  # never lose the line over the extra leg
            out["long_reads"] = {"error": repr(e)}
        call = HostCall(hmm, hc_regions, False)
        out["hc_2000_regions"] = leg(call, 8, f"HaplotypeCaller-shaped stream, {len(hc_regions)} active regions (30x, 121-151 bp reads, 2-32 "
                                              "haplotypes of 250-500 bp) in one compute_regions call")
        out["hc_2000_regions"]["us_per_region"] = round(out["hc_2000_regions"]["wall_ms"] * 1e3 / len(hc_regions), 2)
        # one region per call: how the reference's likelihood engine calls a PairHMM (PairHMM.java:190, once per region and sample)
        calls = [HostCall(hmm, [b], True) for b in hc_regions[:300]]
        for c in calls[:20]:
            c()
        lat = []
        t_all = time.perf_counter()
        for c in calls:
            t0 = time.perf_counter()
            c()
            lat.append(time.perf_counter() - t0)
        t_all = time.perf_counter() - t_all
        cells = sum(c.cells for c in calls)
        out["one_region_per_call"] = {"workload": "300 of those regions, one compute_likelihoods call each, one caller thread",
                                      "median_us": round(float(np.median(lat)) * 1e6, 1), "p90_us": round(float(np.percentile(lat, 90)) * 1e6, 1),
                                      "gcups": round(cells / t_all / 1e9, 1), "roofline_frac": roof_frac(cells, t_all),
                                      "mean_pairs_per_call": round(sum(c.pairs for c in calls) / len(calls), 1)}
        calls = None
        out["one_region_per_call_two_in_flight"] = in_flight_leg(hmm, hc_regions[:300], 2)
        out["one_region_per_call_four_in_flight"] = in_flight_leg(hmm, hc_regions[:300], 4)
        big = max(hc_regions[:300], key=lambda b: b.num_reads)
        out["eight_sample_region"] = region_hook_leg(hmm, big, 8)
    finally:
        hmm.close()
    out["callers_sharing_one_gpu"] = shared_device_legs()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=4096)
    ap.add_argument("--haps", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other workloads (profiling runs)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the host-pointer leg (profiling runs: its calls are cut into parts, whose smaller launches of the same kernel would mix into the kernel's average)")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: whatever libraries print while they set up (gloo announces
    # its connections there) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
    host_group = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            host_group = dist.new_group(backend="gloo")  # the host gather of the c4 leg: CPU arrays, no device collective
        else:
            dist.init_process_group(backend)
            host_group = dist.group.WORLD

    from gatk_amd import sharding, synth
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

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    # ---- resident leg: inputs in HBM when the clock starts ----------------------------------------------------
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
    elapsed = max_over_ranks(elapsed)
    total_cells = sum_over_ranks(batch.cells)

    # sanity: results of the last step are finite log10 likelihoods <= 0
    out = res.download()[0]
    assert out.shape == (batch.num_reads, batch.num_haps) and bool((out <= 0).all())
    res.free()
    hmm.close()

    # ---- end-to-end leg: host arrays in, host arrays out, on a handle with the default options ------------------
    e2e = None
    if not args.no_end_to_end:
        hmm = HipLoglessPairHMM(local_rank)
        call = HostCall(hmm, [batch], True)
        for _ in range(args.warmup):
            call()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            call()
        barrier()
        e2e = max_over_ranks(time.perf_counter() - t0)
        call.check()
        assert float(np.abs(call.outs[0] - out).max()) <= 1e-5  # the two legs score the same batch
        hmm.close()

    # ---- N > 1: the whole-genome-shaped workload, sharded over the ranks, host gather inside the clock ---------
    c4 = None
    if world > 1:
        base = synth.make_regions(2000, seed=synth.DEFAULT_SEED)
        order = np.concatenate([np.random.default_rng(9).permutation(len(base)) for _ in range(5)])
        regions = [base[i] for i in order]  # 10 000 regions; the same list on every rank
        hmm = HipLoglessPairHMM(local_rank)
        try:
            # the assignment and this rank's marshalled call are made once, as a host that holds its arrays would
            shards = sharding.shard_regions(regions, world)
            prepared = hmm.prepare_regions([regions[i] for i in shards[rank]])
            times = []
            for it in range(4):
                barrier()
                t0 = time.perf_counter()
                gathered = sharding.compute_sharded(regions, lambda _mine: prepared.compute(), rank=rank, world=world,
                                                    gather_group=host_group, shards=shards)
                barrier()
                times.append(max_over_ranks(time.perf_counter() - t0))
            if rank == 0:
                assert len(gathered) == len(regions) and all(g.shape == (b.num_reads, b.num_haps) and bool((g <= 0).all()) for g, b in zip(gathered, regions))
                cells = sum(b.cells for b in regions)
                c4 = {"workload": "10 000 HaplotypeCaller-shaped regions (five orderings of 2000 generated ones) sharded over the ranks, "
                                  "longest processing time first by cells; results gathered on rank 0's host (gloo)",
                      "scaling": "strong", "cells": cells, "pairs": sum(b.num_reads * b.num_haps for b in regions),
                      "wall_ms": round(min(times[1:]) * 1e3, 2), "first_ms": round(times[0] * 1e3, 2),
                      "gcups": round(cells / min(times[1:]) / 1e9, 1), "includes": "planning, H2D, kernels, D2H on every rank, then the host gather of the likelihood arrays on rank 0"}
            # the same with every rank's results copied D2H straight into its slice of one shared-memory segment
            # (the ranks share a host): the gather is the barrier
            try:
                shared = sharding.SharedOutput(regions, shards, rank, world, host_group)
                try:
                    in_place = hmm.prepare_regions([regions[i] for i in shards[rank]], flat=shared.mine)
                    times = []
                    for it in range(4):
                        barrier()
                        t0 = time.perf_counter()
                        views = sharding.compute_sharded(regions, lambda _mine: in_place.compute(), rank=rank, world=world,
                                                         gather_group=host_group, shards=shards, shared=shared)
                        barrier()
                        times.append(max_over_ranks(time.perf_counter() - t0))
                    if rank == 0:
                        same = all(np.array_equal(v, g) for v, g in zip(views, gathered))
                        c4["shared_memory_output"] = {"wall_ms": round(min(times[1:]) * 1e3, 2), "gcups": round(cells / min(times[1:]) / 1e9, 1),
                                                      "identical_to_gathered": bool(same),
                                                      "path": "process per GPU; each rank's D2H scatter lands in its slice of one shared-memory segment, then a barrier"}
                    views = in_place = None
                finally:
                    shared.close()
            except Exception as e:  # never lose the line over the extra leg
                if rank == 0 and c4 is not None:
                    c4["shared_memory_output"] = {"error": repr(e)}
        finally:
            hmm.close()
        # and the same list through ONE process driving every GPU (the multi-device handle a single JVM would hold),
        # while the other ranks wait at a host-side barrier (a device-side one would keep their GPUs spinning)
        if rank == 0:
            try:
                devices = [r % torch.cuda.device_count() for r in range(world)]
                multi = HipLoglessPairHMM(devices)
                mcall = HostCall(multi, regions, False)
                med, best = time_calls(mcall, 3, warm=2)
                multi.close()
                c4["one_process_all_gpus"] = {"wall_ms": round(best * 1e3, 2), "gcups": round(mcall.cells / best / 1e9, 1),
                                              "path": "gatk_pairhmm_create_multi + one compute_regions call: sharded inside the C-ABI, "
                                                      "engines write into the caller's arrays, no gather copy"}
            except Exception as e:  # never lose the line over the extra leg
                c4["one_process_all_gpus"] = {"error": repr(e)}
        dist.barrier(group=host_group)

    if rank == 0:
        steps = max(args.steps, 1)
        kern_s = fp32_ms / steps * 1e-3
        achieved = batch.cells * FLOP_PER_CELL / kern_s / 1e12
        kernel = (f"{'fwd_stream' if st['main_stream_pass'] else 'fwd_direct'}<{st['main_rows_per_lane']}, "
                  f"{EMIT_MODES.get(st['main_emit_mode'], st['main_emit_mode'])}> (fp32 forward pass)")
        default_shape = (args.reads, args.haps) == (4096, 64)
        line = {
            "metric": "PairHMM GCUPS", "value": round(total_cells * steps / elapsed / 1e9, 2), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"large-batch PairHMM: {args.reads} reads x {args.haps} haplotypes, 151 bp / 600 bp, per GPU",
                       "cells_per_gpu_step": batch.cells, "pairs_per_gpu_step": batch.num_reads * batch.num_haps,
                       "value_is": "resident pass: the batch is in HBM when the clock starts; every kernel of a pass is inside it",
                       "parallelism": f"independent batches on {world} GPU(s), host gather, no collective"},
            "resident_gcups": round(total_cells * steps / elapsed / 1e9, 2),
            "end_to_end": None if e2e is None else {
                "gcups": round(total_cells * steps / e2e / 1e9, 2), "ms_per_call": round(e2e / steps * 1e3, 4), "calls_timed": args.steps,
                "path": "gatk_pairhmm_compute_likelihoods on host arrays: checks + planning + H2D + kernels + fp64 reruns + D2H"},
            "roofline": {"bound": "valu", "bound_detail": "fp32 vector ALU (a recurrence: neither HBM- nor MFMA-bound)",
                         "kernel": kernel, "achieved": round(achieved, 2), "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
                         "traffic": PMC_TRAFFIC_BYTES_DEFAULT_WORKLOAD if default_shape else None,
                         "traffic_source": PMC_TRAFFIC_SOURCE if default_shape else None,
                         "flop_per_cell": FLOP_PER_CELL,
                         "flop_per_cell_executed": EXECUTED_FLOP_PER_CELL,
                         "frac_executed": round(achieved / PEAK_FP32_VALU_TFLOPS * EXECUTED_FLOP_PER_CELL / FLOP_PER_CELL, 4),
                         "frac_note": "frac prices the 12 FLOP of the reference's recurrence per cell; the rescaled form the kernel evaluates "
                                      "executes 9 (4 FMA + 1 multiply), so the vector ALUs run at frac_executed of their peak",
                         "kernel_ms": round(fp32_ms / steps, 4),
                         "kernel_gcups": round(batch.cells / kern_s / 1e9, 1)},
            "other_kernels_ms": round(aux_ms / steps, 4),
            "fp64_rerun": {"pairs_per_step": rerun_pairs // steps, "rate": round(rerun_pairs / steps / (batch.num_reads * batch.num_haps), 6),
                           "kernel_ms": round(fp64_ms / steps, 4)},
        }
        if c4 is not None:
            line["c4"] = c4
        if world == 1 and not args.no_configs:
            line["configs"] = other_configs(local_rank, synth.make_regions(2000, seed=synth.DEFAULT_SEED))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batch)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
