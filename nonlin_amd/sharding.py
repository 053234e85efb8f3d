"""Independent problems shard across ranks (one process per GPU); no data-path collective.

A batch of nprob independent LM/Newton problems is embarrassingly parallel:
problem k goes to rank k mod world (block-cyclic, because iteration counts differ per
problem).  Collectives appear only at the two ends: a broadcast of the options / base seed
from rank 0 and a gather of per-problem results.  Works with backend "nccl" (= RCCL over
xGMI on ROCm) and with "gloo" (CPU tests).
"""
import torch
import torch.distributed as dist


def shard_indices(nprob, rank, world):
    """Global indices of the problems rank `rank` owns (block-cyclic)."""
    return list(range(rank, nprob, world))


def shard_count(nprob, rank, world):
    return len(range(rank, nprob, world))


def broadcast_config(values, device, src=0):
    """Broadcast a flat list of floats (options, base seed ...) from rank `src`."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return [float(v) for v in t.tolist()]


def gather_results(local, nprob, rank, world):
    """Gather per-problem result rows.  `local` is [nlocal, width] float64 holding the rows of
    this rank's problems in shard order; returns [nprob, width] in global problem order on
    every rank (all_gather of equal-sized padded shards)."""
    width = local.shape[1]
    if world == 1:
        return local.clone()
    per = (nprob + world - 1) // world
    pad = torch.zeros((per, width), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    full = torch.empty((nprob, width), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(nprob, r, world)
        if idx:
            full[torch.tensor(idx, device=local.device)] = out[r * per: r * per + len(idx)]
    return full


def solve_sharded(nprob, rank, world, device, config, solve_local):
    """One sharded batch from end to end: broadcast `config` (a flat list of numbers, rank 0's values win), deal the
    problems block-cyclically, let `solve_local(config, global_indices)` solve this rank's share and return one float64
    row per problem ([nlocal, width], shard order), gather the rows in global problem order on every rank.

    bench.py's N > 1 path is these three calls around its timed loop; the CPU tests drive the same function over gloo
    with a stub `solve_local`, the GPU tests with the real batched solver.  Returns (config, rows [nprob, width])."""
    cfg = broadcast_config(config, device)
    idx = shard_indices(nprob, rank, world)
    local = solve_local(cfg, idx)
    if local.dim() != 2 or local.shape[0] != len(idx):
        raise ValueError(f"solve_local returned {tuple(local.shape)} for {len(idx)} problems")
    return cfg, gather_results(local, nprob, rank, world)
