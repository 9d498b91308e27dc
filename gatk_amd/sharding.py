"""Sharding of PairHMM work over the GPUs of one node: one process per GPU, no data-path collective.

Every (read, haplotype) pair is independent (PairHMM.java:217,284 force a full recompute per
pair) and so is every active region (TraverseActiveRegions.java:705-709 maps them one by one),
so the unit of distribution is the region batch, or a read range of one very large batch.  Each
rank scores its share on its own GPU; the likelihood arrays are then gathered on the host of
rank 0 (a gloo gather of CPU arrays).  Nothing is exchanged between GPUs.
"""
import numpy as np


def lpt_assign(weights, num_shards):
    """Longest-processing-time-first assignment: returns num_shards lists of item indices with
    nearly equal weight sums.  Deterministic (ties broken by index)."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0] * num_shards
    shards = [[] for _ in range(num_shards)]
    for i in order:
        k = min(range(num_shards), key=lambda s: (loads[s], s))
        shards[k].append(i)
        loads[k] += weights[i]
    for s in shards:
        s.sort()
    return shards


def shard_regions(regions, num_shards):
    """Region indices per shard, balanced by cell count (sum readLen * hapLen)."""
    return lpt_assign([b.cells for b in regions], num_shards)


def split_reads(batch, num_shards):
    """Read ranges [lo, hi) that split one large batch into num_shards parts of nearly equal read
    bases; every part carries all haplotypes."""
    if batch.num_reads == 0:
        return [(0, 0)] * num_shards
    cum = np.asarray(batch.read_offsets[1:], dtype=np.int64)
    total = int(cum[-1])
    cuts = [0]
    for k in range(1, num_shards):
        cuts.append(int(np.searchsorted(cum, total * k / num_shards, side="left")) + 1 if total else 0)
    cuts.append(batch.num_reads)
    cuts = [min(max(c, 0), batch.num_reads) for c in cuts]
    for k in range(1, len(cuts)):
        cuts[k] = max(cuts[k], cuts[k - 1])
    return [(cuts[k], cuts[k + 1]) for k in range(num_shards)]


def compute_sharded(regions, compute_fn, rank=0, world=1, gather_group=None):
    """Score `regions` (list of Batch) with `compute_fn(list_of_batches) -> list of [reads, haps]
    arrays` on this rank's share and gather on rank 0.

    Returns the per-region arrays in input order on rank 0 and None on the other ranks.  With
    world == 1 no process group is touched.  gather_group must be a CPU (gloo) group: the gather
    is a host-side copy, not a GPU collective."""
    shards = shard_regions(regions, world)
    mine = shards[rank]
    outs = compute_fn([regions[i] for i in mine]) if mine else []
    if world == 1:
        return list(outs)
    import torch.distributed as dist

    payload = [(i, np.ascontiguousarray(o)) for i, o in zip(mine, outs)]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(payload, gathered, dst=0, group=gather_group)
    if rank != 0:
        return None
    result = [None] * len(regions)
    for part in gathered:
        for i, o in part:
            result[i] = o
    return result


def compute_split_batch(batch, compute_fn, rank=0, world=1, gather_group=None):
    """One large batch split by read ranges over the ranks; rank 0 gets the [reads, haps] array."""
    ranges = split_reads(batch, world)
    lo, hi = ranges[rank]
    part = compute_fn([batch.read_slice(lo, hi)])[0] if hi > lo else np.zeros((0, batch.num_haps))
    if world == 1:
        return part
    import torch.distributed as dist

    gathered = [None] * world if rank == 0 else None
    dist.gather_object(np.ascontiguousarray(part), gathered, dst=0, group=gather_group)
    if rank != 0:
        return None
    return np.concatenate([g for g in gathered if g.shape[0] > 0], axis=0) if batch.num_reads else part
