"""Multi-GPU layout of the sampling path: one process per GPU, batch-sharded, NO collective in the loop.

Samples (and long-audio chunks) are independent and the weights replicate per GPU, so the
only exchange is the final gather of the generated latents.  With RCCL (`backend="nccl"`) the gather runs
over xGMI; the same code runs under `gloo` on CPU tensors, which is how tests/test_dist_cpu.py covers it.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int):
    """Contiguous [start, end) of `total` items for `rank`: sizes differ by at most one
    (28 items over 8 ranks -> 4,4,4,4,3,3,3,3)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def sample_sharded(sample_fn, lr_latent, z0, group=None):
    """Run `sample_fn(lr_slice, z0_slice) -> z_slice` on this rank's slice of the batch and return the full
    [B, C, T] result on every rank (all_gather of equal-size padded slices)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = lr_latent.shape[0]
    a, b = shard_range(B, world, rank)
    local = sample_fn(lr_latent[a:b], z0[a:b]) if b > a else lr_latent[:0].clone()
    if world == 1:
        return local
    width = -(-B // world)
    pad = torch.zeros((width,) + tuple(lr_latent.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: b - a] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r in range(world):
        ra, rb = shard_range(B, world, r)
        out.append(parts[r][: rb - ra])
    return torch.cat(out, 0)


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX all-reduce of a host scalar (bench timing contract)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sample_long_sharded(sample_chunks_fn, plan, group=None):
    """BASELINE configs[4] layout: the chunks of one long file (reference chunk plan, infer_test_v3m2.py:340-361)
    are independent, so rank r samples chunks r, r + world, ... with `sample_chunks_fn(indices) -> {index: tensor}`
    and every rank receives all generated chunks in plan order (ready for `crossfade_chunks`).  Chunk lengths
    differ (the last one is shorter), so the exchange is an object gather of CPU tensors — it happens once per
    file, outside the sampling loop, which has no collective."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = list(range(rank, len(plan), world))
    local = {i: t.detach().cpu() for i, t in sample_chunks_fn(mine).items()} if mine else {}
    if world == 1:
        return [local[i] for i in range(len(plan))]
    parts = [None] * world
    dist.all_gather_object(parts, local, group=group)
    merged = {}
    for p in parts:
        merged.update(p)
    return [merged[i] for i in range(len(plan))]
