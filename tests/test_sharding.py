"""CPU tests of the N > 1 path: block-cyclic sharding of independent problems and the
broadcast / gather ends, world_size 2 over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nonlin_amd import sharding


def test_shard_indices_cover_everything_once():
    for nprob in (1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                idx = sharding.shard_indices(nprob, r, world)
                assert len(idx) == sharding.shard_count(nprob, r, world)
                seen += idx
            assert sorted(seen) == list(range(nprob))
    assert sharding.shard_indices(10, 1, 4) == [1, 5, 9]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nprob, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = sharding.broadcast_config([500, 12345, 0.5] if rank == 0 else [0, 0, 0], torch.device("cpu"))
        idx = sharding.shard_indices(nprob, rank, world)
        # each rank "solves" its problems: row = [global index, seed, 2*index]
        local = torch.tensor([[float(k), cfg[1] + k, 2.0 * k] for k in idx], dtype=torch.float64).reshape(len(idx), 3)
        full = sharding.gather_results(local, nprob, rank, world)
        q.put((rank, cfg, full.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nprob", [5, 8])
def test_broadcast_and_gather_world2_gloo(nprob):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nprob, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, cfg, full in out:
        assert cfg == [500.0, 12345.0, 0.5]                   # rank 0's values reached everyone
        assert full == [[float(k), 12345.0 + k, 2.0 * k] for k in range(nprob)]   # global order restored
