"""world_size-2 gloo test of the frame sharding + table gather used by bench.py for N > 1."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pli_slam_amd.sharding import TableGatherer, exchange_halo, gather_tables, shard_range


def test_shard_range_partitions_the_batch():
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s1 == s0 + c0
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_range(256, 3, 8) == (96, 32)          # config 4: 256-frame batch, 32 per GPU


def _worker(rank, world, path, nframes, rec):
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=world)
    start, count = shard_range(nframes, rank, world)
    # a table whose bytes encode (global frame index, byte offset) so misplaced records are detected
    t = torch.zeros(count * rec, dtype=torch.uint8)
    for f in range(count):
        t[f * rec:(f + 1) * rec] = torch.from_numpy(((np.arange(rec) + 31 * (start + f)) % 251).astype(np.uint8))
    out = gather_tables(t, rec, count, dst=0)
    known = gather_tables(t, rec, count, dst=0, counts=[shard_range(nframes, r, world)[1] for r in range(world)])
    assert (out is None) == (known is None)
    if out is not None:
        assert all(torch.equal(a, b) for a, b in zip(out, known))      # counts known in advance: same result, no exchange
    if rank == 0:
        assert len(out) == world
        full = torch.cat(out)
        assert full.numel() == nframes * rec
        for f in range(nframes):
            want = ((np.arange(rec) + 31 * f) % 251).astype(np.uint8)
            assert np.array_equal(full[f * rec:(f + 1) * rec].numpy(), want), f
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nframes", [8, 7])
def test_gather_tables_gloo_world2(nframes):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "rdv"), nframes, 96), nprocs=2, join=True)


def _worker_async(rank, world, path, steps, nbytes):
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=world)
    g = TableGatherer(nbytes, torch.device("cpu"), depth=2, dst=0)
    seen = {}
    for i in range(steps):
        slot = g.acquire()
        if rank == 0 and i >= 2:                     # the gather that used this slot two steps ago is complete here
            seen[i - 2] = [b.clone() for b in g.gathered(slot)]
        g.table(slot).copy_(torch.full((nbytes,), (17 * i + 3 * rank) % 251, dtype=torch.uint8))   # "the kernels of step i"
        g.submit(slot)
    g.drain()
    if rank == 0:
        for i in range(max(0, steps - 2), steps):
            seen[i] = [b.clone() for b in g.gathered(i % 2)]
        for i in range(steps):
            for r in range(world):
                assert int(seen[i][r][0]) == (17 * i + 3 * r) % 251 and bool((seen[i][r] == seen[i][r][0]).all()), (i, r)
    dist.barrier()
    dist.destroy_process_group()


def test_async_double_buffered_gather_gloo_world2():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_async, args=(2, os.path.join(d, "rdv"), 5, 4096), nprocs=2, join=True)


def _worker_halo(rank, world, path, nframes, rec):
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=world)
    start, count = shard_range(nframes, rank, world)
    t = torch.zeros((count + 1) * rec, dtype=torch.uint8)                    # [halo | local frames]
    for f in range(count):
        t[(f + 1) * rec:(f + 2) * rec] = (start + f) % 251
    got = exchange_halo(t, rec, count)
    if rank == 0 or shard_range(nframes, rank - 1, world)[1] == 0:
        assert not got and int(t[:rec].max()) == 0                           # nothing in front of the stream's first frame
    else:
        assert got and bool((t[:rec] == (start - 1) % 251).all())            # the frame before this shard's first one
    for f in range(count):                                                   # the local records are untouched
        assert bool((t[(f + 1) * rec:(f + 2) * rec] == (start + f) % 251).all())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nframes", [(2, 8), (3, 7), (3, 2), (8, 256), (8, 7)])
def test_one_frame_halo_across_shard_borders_gloo(world, nframes):
    """rank r receives the table record of frame start_r - 1 (f2f matching across shard borders).  (8, 256): the
    partition of BASELINE configs[3] (32 frames per rank); (8, 7): eight ranks, the last one without frames."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_halo, args=(world, os.path.join(d, "rdv"), nframes, 128), nprocs=world, join=True)
