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


def exchange_sum_(buf, group=None, async_op=False):
    """Sum the contiguous 1-D tensor `buf` over the ranks of `group`, in place on every rank — the gradient exchange of the
    training step (the reference's DDP all-reduce, train_ddp_v3mod2.py:822,922) as reduce-scatter + all-gather:

        rank r reduces shard r (buf[r*n/W : (r+1)*n/W]) from all ranks, then every rank gathers the W reduced shards.

    On an 8-GPU MI355X node the GPUs are fully connected by xGMI (7 links x ~153 GB/s per GPU): both phases keep all 7
    links of every GPU busy with one direct transfer per peer (2 * 7/8 * n bytes per GPU in total, ~1/7 of it per link),
    where a ring all-reduce is bound by ONE link per direction (~5 ms vs ~35 ms for the 3.06 GB of fp32
    gradients).  Both collectives are in place (shard r of the output aliases the input, the NCCL/RCCL in-place form).
    Falls back to one all_reduce when the length is not divisible by the world size, and for slices below
    `JAT_EXCHANGE_MIN_BYTES` (default 1 MiB: two collectives' launch latency outweighs the link argument there);
    JAT_EXCHANGE_ALLREDUCE=1 forces plain all_reduce everywhere (A/B on hardware).  Returns the list of work handles
    (async_op=True) to wait on, in order.  UNMEASURED on hardware: no multi-GPU node was available to the build — the
    5 ms / 35 ms figures are a link-count model, not a measurement."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return []
    n = buf.numel()
    if buf.dim() != 1 or not buf.is_contiguous():
        raise ValueError("exchange_sum_ needs a contiguous 1-D tensor")
    import os
    small = n * buf.element_size() < int(os.environ.get("JAT_EXCHANGE_MIN_BYTES", 1 << 20))
    if n % world != 0 or small or os.environ.get("JAT_EXCHANGE_ALLREDUCE") == "1":
        w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return [w] if async_op else []
    rank = dist.get_rank(group)
    shard = n // world
    mine = buf[rank * shard:(rank + 1) * shard]
    w1 = dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    if async_op:
        # The all-gather reads the shard the reduce-scatter writes: order them EXPLICITLY on every backend instead of relying on
        # the backend's own queueing.  With RCCL `wait()` makes the CURRENT stream (the caller's communication stream, see
        # Trainer._on_grads_ready) wait for the collective's completion event — the host does not block; with gloo it joins the
        # worker thread.  One code path: the branch an 8-GPU run takes is the branch the world_size-2 gloo test takes.
        w1.wait()
    w2 = dist.all_gather_into_tensor(buf, mine, group=group, async_op=async_op)
    return [w1, w2] if async_op else []
