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


def _worker_solve(rank, world, port, nprob, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []

        def solve_local(cfg, idx):
            # stand-in for DeviceSolver.lm_solve_batch on this rank's share: problem k "converges" in 4 + k % 3
            # iterations and its checksum depends on the broadcast seed -- what bench.py gathers per problem
            seen.extend(idx)
            seed0 = cfg[1]
            return torch.tensor([[4.0 + k % 3, 5.0 + k % 3, 3.0 + k % 3, seed0 + k] for k in idx],
                                dtype=torch.float64).reshape(len(idx), 4)
        cfg, rows = sharding.solve_sharded(nprob, rank, world, torch.device("cpu"),
                                           [500, 12345, 0.5] if rank == 0 else [1, 2, 3], solve_local)
        q.put((rank, cfg, seen, rows.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nprob", [7, 16])
def test_solve_sharded_world2_gloo(nprob):
    """shard -> solve -> gather through the function bench.py's multi-GPU path is built from."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_solve, args=(r, world, port, nprob, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [[4.0 + k % 3, 5.0 + k % 3, 3.0 + k % 3, 12345.0 + k] for k in range(nprob)]
    solved = []
    for rank, cfg, seen, rows in out:
        assert cfg == [500.0, 12345.0, 0.5]
        assert seen == list(range(rank, nprob, world))        # block-cyclic share
        assert rows == expect                                  # every rank holds all results in global order
        solved += seen
    assert sorted(solved) == list(range(nprob))                # every problem solved exactly once


def test_bench_gpus_flag_must_match_world_size():
    """bench.py --gpus N under a launcher that started a different number of ranks fails before touching a GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
