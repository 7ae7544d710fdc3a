"""N>1 path on CPU: world_size-2 gloo processes shard the batch, sample their slice (the sampler here is the
numpy oracle standing in for the GPU kernels) and gather — result must equal the unsharded run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jatsr_amd.recipe as recipe
    from jatsr_amd.dist import max_over_ranks, sample_sharded
    from oracle import jat_oracle as O
    cfg = recipe.CONFIGS["micro"]
    orc = O.OracleModel(cfg, recipe.make_state_dict(cfg, threads=1))
    B = 3                                               # odd: ranks get 2 and 1 samples
    lr = torch.from_numpy(recipe.gaussian("lr_latent", (B, 32, 16), 200))
    z0 = torch.from_numpy(recipe.gaussian("z0", (B, 32, 16), 201))

    def fn(lr_s, z_s):
        return torch.from_numpy(O.flow_matching_sample(orc, lr_s.numpy(), z_s.numpy(), 3, 2.0))

    full = sample_sharded(fn, lr, z0)
    t = max_over_ranks(1.0 + rank)
    # config-5 layout: chunks of one long file sharded over ranks, gathered in plan order, crossfaded
    from jatsr_amd.dist import sample_long_sharded
    plan = O.chunk_plan(100, 40, 8)
    lr_long = recipe.gaussian("long_lr", (1, 32, 100), 5)
    noise = [recipe.gaussian("noise", (1, 32, b - a), i) for i, (a, b) in enumerate(plan)]

    def chunk_fn(idx):
        return {i: torch.from_numpy(O.flow_matching_sample(orc, lr_long[:, :, plan[i][0]:plan[i][1]], noise[i], 2, 2.0))
                for i in idx}

    chunks = sample_long_sharded(chunk_fn, plan)
    long_out = O.crossfade_chunks([c.numpy() for c in chunks], 8)
    if rank == 0:
        q.put((full.numpy(), t, long_out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sampling_equals_unsharded_gloo_world2():
    import jatsr_amd.recipe as recipe
    from oracle import jat_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, tmax, long_out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = recipe.CONFIGS["micro"]
    orc = O.OracleModel(cfg, recipe.make_state_dict(cfg, threads=1))
    lr = recipe.gaussian("lr_latent", (3, 32, 16), 200)
    z0 = recipe.gaussian("z0", (3, 32, 16), 201)
    ref = O.flow_matching_sample(orc, lr, z0, 3, 2.0)
    assert got.shape == ref.shape and np.allclose(got, ref, atol=1e-5)
    assert tmax == 2.0
    plan = O.chunk_plan(100, 40, 8)
    lr_long = recipe.gaussian("long_lr", (1, 32, 100), 5)
    noise = [recipe.gaussian("noise", (1, 32, b - a), i) for i, (a, b) in enumerate(plan)]
    ref_long = O.crossfade_chunks([O.flow_matching_sample(orc, lr_long[:, :, a:b], noise[i], 2, 2.0)
                                   for i, (a, b) in enumerate(plan)], 8)
    assert long_out.shape == (1, 32, 100) and np.allclose(long_out, ref_long, atol=1e-5)
