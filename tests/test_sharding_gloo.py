"""The N > 1 path on CPU: two processes (gloo) each score their share of a region stream and of
one large batch, rank 0 gathers on the host (gloo gather, and the shared-memory output of a single node).  The per-rank compute is the CPU oracle here (no
GPU in this tier); on the GPU box the same functions are driven with the HIP engine
(tests/test_gpu_parity.py::test_sharded_regions_single_rank)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from gatk_amd import sharding, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_compute(batches):
    from oracle import oracle as O
    outs = []
    for b in batches:
        rc, o = O.compute_likelihoods(b, O.LOGLESS, fast_threads=1)
        assert rc == 0
        outs.append(o)
    return outs


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        regions = synth.make_regions(9, seed=21, coverage=6, read_len=60)
        got = sharding.compute_sharded(regions, _oracle_compute, rank, world)
        big = synth.make_batch(23, 3, 40, 90, seed=22, len_jitter=15)
        whole = sharding.compute_split_batch(big, _oracle_compute, rank, world)
        # the same list through the shared-memory output of a single node: the gather is a barrier
        shards = sharding.shard_regions(regions, world)
        shared = sharding.SharedOutput(regions, shards, rank, world)
        try:
            via_shm = sharding.compute_sharded(regions, _oracle_compute, rank, world, shards=shards, shared=shared)
            via_shm = [np.array(v) for v in via_shm] if rank == 0 else None  # copies: the segment goes away below
        finally:
            shared.close()
        if rank == 0:
            q.put((got, whole, via_shm))
        else:
            assert got is None and whole is None and via_shm is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, whole, via_shm = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    regions = synth.make_regions(9, seed=21, coverage=6, read_len=60)
    want = _oracle_compute(regions)
    assert len(got) == len(want)
    for g, w, v in zip(got, want, via_shm):
        np.testing.assert_array_equal(g, w)
        np.testing.assert_array_equal(v, w)
    big = synth.make_batch(23, 3, 40, 90, seed=22, len_jitter=15)
    np.testing.assert_array_equal(whole, _oracle_compute([big])[0])
    # both ranks had work
    shards = sharding.shard_regions(regions, world)
    assert all(len(s) > 0 for s in shards)
