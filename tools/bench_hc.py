#!/usr/bin/env python3
"""Wall time of a HaplotypeCaller-shaped stream of active regions through the host-pointer entry point.

The walker that would produce these batches is not in the reference tree, so the stream is synthetic
(gatk_amd/synth.make_regions: region span 50-300 bp + 100 bp padding either side, 2-32 haplotypes,
reads = coverage x span / 151, read length 121-151).  What is timed is what a caller that holds
host arrays pays at the C-ABI: checking, planning, one H2D copy, the kernels, one D2H copy per chunk
of regions (gatk_pairhmm_compute_regions).  Worker threads, each with its own engine handle and stream, take chunks in turn, so one
chunk's host work overlaps another chunk's kernels (the reference creates one PairHMM per map thread,
pairhmm/PairHMM.java:297).

With WORLD_SIZE > 1 (torchrun) every rank scores its share of the regions (longest-processing-time
assignment by cell count) on its own GPU and rank 0 gathers the arrays on the host with gloo.

Prints one JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatk_amd import pairhmm, sharding, synth  # noqa: E402
from gatk_amd.pairhmm import HipLoglessPairHMM  # noqa: E402


def score(regions, device, chunk, workers, combine=False, options=()):
    """Scores regions in chunks on `workers` threads; returns (outs, wall seconds).  The ctypes views of
    the caller's arrays and the output arrays are made beforehand: the clock covers the C-ABI calls."""
    lib = pairhmm.load_library()
    chunks = [list(range(k, min(k + chunk, len(regions)))) for k in range(0, len(regions), chunk)]
    outs = [np.full((b.num_reads, b.num_haps), np.nan, dtype=np.float64) for b in regions]
    keep, calls = [], []
    for idx in chunks:
        cbs = (pairhmm.CBatch * len(idx))(*[pairhmm.to_cbatch(regions[i], keep) for i in idx])
        arr = (C.c_void_p * len(idx))(*[outs[i].ctypes.data if outs[i].size else None for i in idx])
        calls.append((len(idx), cbs, arr))
    handles = [HipLoglessPairHMM(device) for _ in range(workers)]
    for h in handles:
        for name, value in options:
            h.set_option(name, value)
    if combine:  # region-sized calls of all workers share device passes (handle option combine_calls)
        for h in handles:
            h.set_option("combine_calls", 1)
    for h in handles:  # an engine in steady state: buffers grown to the largest of these calls, clocks up
        for n, cbs, arr in (calls if workers > 1 else calls[:1] * 3):
            assert lib.gatk_pairhmm_compute_regions(h._h, n, cbs, arr) == 0
    nxt = [0]
    lock = threading.Lock()
    errors = []

    def work(h):
        while True:
            with lock:
                k = nxt[0]
                nxt[0] += 1
            if k >= len(calls):
                return
            n, cbs, arr = calls[k]
            if combine and n == 1:
                rc = lib.gatk_pairhmm_compute_likelihoods(h._h, C.byref(cbs[0]), arr[0])
            else:
                rc = lib.gatk_pairhmm_compute_regions(h._h, n, cbs, arr)
            if rc != 0:
                errors.append(RuntimeError(lib.gatk_pairhmm_last_error().decode()))
                return

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(h,)) for h in handles]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    last = handles[0].stats()
    for h in handles:
        h.close()
    if errors:
        raise errors[0]
    return outs, wall, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regions", type=int, default=2000)
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--chunk", type=int, default=250, help="regions per engine call")
    ap.add_argument("--workers", type=int, default=3, help="host threads, one engine handle each")
    ap.add_argument("--default-quals", action="store_true",
                    help="pass no insertion / deletion / GCP arrays: the engine's defaults (Q45, Q45, 10), which is what these synthetic reads carry anyway")
    ap.add_argument("--cpu", action="store_true", help="also time the vectorised CPU baseline on the same regions")
    ap.add_argument("--combine", action="store_true", help="with --chunk 1: one compute_likelihoods call per region with the combine_calls option")
    ap.add_argument("--option", action="append", default=[], help="engine option as name=value (repeatable)")
    ap.add_argument("--check", type=int, default=8, help="regions compared with the oracle afterwards")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    group = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")  # host gather only: no device collective on this path
        group = dist.group.WORLD

    regions = synth.make_regions(args.regions, seed=5, coverage=args.coverage)
    cells = sum(b.cells for b in regions)
    full_regions = regions
    if args.default_quals:
        from gatk_amd.batch import Batch
        regions = [Batch(b.read_bases, b.read_quals, None, None, None, b.read_offsets, b.hap_bases, b.hap_offsets) for b in regions]
    mine = sharding.shard_regions(regions, world)[rank]
    my_regions = [regions[i] for i in mine]
    if group is not None:
        dist.barrier()
    outs, wall, last = score(my_regions, local_rank, args.chunk, args.workers, args.combine, [(o.split("=")[0], float(o.split("=")[1])) for o in args.option])
    if group is not None:
        t = [None] * world
        dist.all_gather_object(t, wall)
        wall = max(t)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((mine, outs), gathered, dst=0)
        if rank == 0:
            full = [None] * len(regions)
            for idx, o in gathered:
                for i, x in zip(idx, o):
                    full[i] = x
            outs = full
    if rank != 0:
        dist.destroy_process_group()
        return
    line = {
        "workload": f"{args.regions} active regions, {args.coverage}x, 151 bp reads (synthetic HaplotypeCaller batch shapes)",
        "reads": int(sum(b.num_reads for b in regions)), "pairs": int(sum(b.num_reads * b.num_haps for b in regions)),
        "cells": int(cells), "n_gpus": world, "default_indel_and_gcp_arrays": bool(args.default_quals), "chunk_regions": args.chunk, "workers_per_gpu": args.workers, "combine_calls": bool(args.combine),
        "wall_ms": round(wall * 1e3, 2), "gcups_including_host_and_pcie": round(cells / wall / 1e9, 1),
        "last_call_of_worker_0": {"host_plan_ms": round(last["host_plan_ms"], 2), "host_upload_ms": round(last["host_upload_ms"], 2)},
    }
    if args.check:
        from oracle import oracle as O
        worst = 0.0
        for i in np.linspace(0, len(regions) - 1, args.check).astype(int):
            rc, want = O.compute_likelihoods(full_regions[i], O.LOGLESS, fast_threads=8)
            assert rc == 0
            fin = np.isfinite(want)
            worst = max(worst, float(np.abs(outs[i][fin] - want[fin]).max()))
        assert worst <= 1e-3, worst
        line["max_abs_err_vs_oracle_on_checked_regions"] = worst
    if args.cpu:
        from oracle import oracle as O
        threads = max(1, min(16, os.cpu_count() or 1))
        t0 = time.perf_counter()
        for b in full_regions:
            assert O.simd_compute_likelihoods(b, threads)[0] == 0
        dt = time.perf_counter() - t0
        line["cpu_simd_baseline"] = {"wall_ms": round(dt * 1e3, 1), "gcups": round(cells / dt / 1e9, 2), "cores": threads, "isa": O.simd_isa()}
    print(json.dumps(line), flush=True)
    if group is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
