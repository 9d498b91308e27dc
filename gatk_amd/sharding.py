"""Sharding of PairHMM work over the GPUs of one node: one process per GPU, no data-path collective.

Every (read, haplotype) pair is independent (PairHMM.java:217,284 force a full recompute per
pair) and so is every active region (TraverseActiveRegions.java:705-709 maps them one by one),
so the unit of distribution is the region batch, or a read range of one very large batch.  Each
rank scores its share on its own GPU; the likelihood arrays are then gathered on the host of
rank 0: a gloo gather of CPU arrays, or - the ranks of one node - every rank's results written straight into its
slice of one shared-memory segment (SharedOutput), where the gather is a barrier.  Nothing is exchanged between GPUs.  A host that is one process
uses the multi-device handle of the C-ABI instead (gatk_pairhmm_create_multi), which shards the same way
inside the library and needs no gather.
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


def _gather_flat(flat, sizes, rank, world, group):
    """Host-side gather of one float64 array per rank on rank 0: every rank knows every size (the region list and the
    assignment are the same everywhere), so the arrays travel as plain buffers padded to the largest, without
    pickling.  Returns the list of per-rank arrays on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    width = max(max(sizes), 1)
    send = torch.zeros(width, dtype=torch.float64)
    if flat.size:
        send[:flat.size] = torch.from_numpy(np.ascontiguousarray(flat))
    recv = [torch.empty(width, dtype=torch.float64) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0, group=group)
    if rank != 0:
        return None
    return [recv[r].numpy()[:sizes[r]] for r in range(world)]


class SharedOutput:
    """One shared-memory segment for the likelihoods of a whole region list, for ranks that run on one host: rank r's
    share of the list (shards[r], in that order) occupies a contiguous slice, `mine` is this rank's, and a
    compute_sharded call that is given the object only has to wait at a barrier: every device's results are copied D2H
    into their place in the output, and nothing is gathered afterwards.  Collective: every rank of `group` (a gloo
    group) constructs it with the same regions and shards."""

    def __init__(self, regions, shards, rank, world, group=None):
        from multiprocessing import resource_tracker, shared_memory

        import torch.distributed as dist

        self.rank, self.world, self.group = rank, world, group
        self.sizes = [sum(regions[i].num_reads * regions[i].num_haps for i in s) for s in shards]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        total = int(self.offsets[-1])
        name = [None]
        if rank == 0:
            self._shm = shared_memory.SharedMemory(create=True, size=max(8, total * 8))
            name[0] = self._shm.name
        dist.broadcast_object_list(name, src=0, group=group)
        if rank != 0:
            self._shm = shared_memory.SharedMemory(name=name[0])
            # the segment belongs to rank 0: this process's resource tracker must not remove it at exit
            try:
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        self.flat = np.ndarray((total,), dtype=np.float64, buffer=self._shm.buf)
        self.mine = self.flat[int(self.offsets[rank]):int(self.offsets[rank + 1])]
        dist.barrier(group=group)

    def views(self, regions, shards):
        """The per-region [reads, haps] views in input order (meaningful on every rank once all have written)."""
        result = [None] * len(regions)
        for r in range(self.world):
            off = int(self.offsets[r])
            for i in shards[r]:
                b = regions[i]
                n = b.num_reads * b.num_haps
                result[i] = self.flat[off:off + n].reshape(b.num_reads, b.num_haps)
                off += n
        return result

    def close(self):
        """Collective: no rank may still hold views into the segment."""
        import torch.distributed as dist

        dist.barrier(group=self.group)
        self.flat = self.mine = None
        self._shm.close()
        if self.rank == 0:
            self._shm.unlink()


def compute_sharded(regions, compute_fn, rank=0, world=1, gather_group=None, shards=None, shared=None):
    """Score `regions` (list of Batch) with `compute_fn(list_of_batches) -> list of [reads, haps]
    arrays` on this rank's share and gather on rank 0.

    Returns the per-region arrays in input order on rank 0 and None on the other ranks.  With
    world == 1 no process group is touched.  gather_group must be a CPU (gloo) group: the gather
    is a host-side copy, not a GPU collective.  `shards` (the result of shard_regions for this list and
    world) may be passed by a caller that scores the same list more than once.  A compute_fn whose result
    carries the arrays as consecutive views of one buffer (attribute `flat_buffer`) is gathered without a copy.
    `shared` (a SharedOutput made for this list and these shards): the results go into this rank's slice of the shared
    segment - nothing to do when compute_fn scored into it already - and the gather is a barrier."""
    if shards is None:
        shards = shard_regions(regions, world)
    mine = shards[rank]
    outs = compute_fn([regions[i] for i in mine]) if mine else []
    if world == 1:
        return list(outs)
    flat = getattr(outs, "flat_buffer", None)
    if shared is not None:
        import torch.distributed as dist

        if flat is None or flat.size == 0 or flat.__array_interface__["data"][0] != shared.mine.__array_interface__["data"][0]:
            off = 0
            for o in outs:
                o = np.asarray(o, dtype=np.float64).reshape(-1)
                shared.mine[off:off + o.size] = o
                off += o.size
            assert off == shared.mine.size
        dist.barrier(group=gather_group)
        return shared.views(regions, shards) if rank == 0 else None
    if flat is None:
        flat = np.concatenate([np.asarray(o, dtype=np.float64).reshape(-1) for o in outs]) if outs else np.zeros(0)
    sizes = [sum(regions[i].num_reads * regions[i].num_haps for i in s) for s in shards]
    assert flat.size == sizes[rank]
    parts = _gather_flat(flat, sizes, rank, world, gather_group)
    if rank != 0:
        return None
    result = [None] * len(regions)
    for r in range(world):
        off = 0
        for i in shards[r]:
            b = regions[i]
            n = b.num_reads * b.num_haps
            result[i] = parts[r][off:off + n].reshape(b.num_reads, b.num_haps)  # a view of the gathered buffer
            off += n
    return result


def compute_split_batch(batch, compute_fn, rank=0, world=1, gather_group=None):
    """One large batch split by read ranges over the ranks; rank 0 gets the [reads, haps] array."""
    ranges = split_reads(batch, world)
    lo, hi = ranges[rank]
    part = compute_fn([batch.read_slice(lo, hi)])[0] if hi > lo else np.zeros((0, batch.num_haps))
    if world == 1:
        return part
    sizes = [(hi_ - lo_) * batch.num_haps for lo_, hi_ in ranges]
    parts = _gather_flat(np.ascontiguousarray(part, dtype=np.float64).reshape(-1), sizes, rank, world, gather_group)
    if rank != 0:
        return None
    return np.concatenate(parts).reshape(batch.num_reads, batch.num_haps)
