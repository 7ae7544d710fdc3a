#!/usr/bin/env python3
"""bench.py — DiT latent-frames/s on MI355X (BASELINE.json metric).

One "step" = one full 50-step CFG=3.0 flow-matching sampling run (hipGraph replay) of a [B=28, C=1024,
T=512] batch of synthetic latents through the v3mod2 DiT (BASELINE.json configs[2], the configuration the
metric string is quoted on; it fits one GPU).  value = output latent frames per second, whole job:
N_gpus * B * T * steps / wall (weak scaling: every rank samples its own B=28 batch; the path has no
data-path collective).  Inputs are resident in HBM before the timed region starts.

Extra objects on the same JSON line:
  roofline     — the dominant kernel (MLP fc1 bf16 MFMA GEMM, M=2B*N_tok, N=5120, K=1280, GELU epilogue):
                 algorithmic FLOPs per launch / average launch duration measured live with HIP events.
  cpu_baseline — the numpy oracle (port of the reference's fp32 CPU forward) timed on this host's cores on a
                 bounded sample, converted to the same unit.
  forward      — single DiT forward (configs[1]) frames/s and its fraction of the bf16 MFMA peak.

    python bench.py                                    # 1 GPU, defaults
    python bench.py --gpus 8 [--scaling strong]        # starts 8 ranks itself (torch.distributed.run child, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          # the driver's own launch: used as is

--scaling weak (default): every rank samples its own B=28 batch.  --scaling strong: ONE B=28 batch is sharded over the
ranks with jatsr_amd.dist.shard_range (28 -> 14,14 -> 7,7,7,7 -> 4,4,4,4,3,3,3,3), value = 28*T*steps / max-over-ranks time.
--dry-run: CPU-only rehearsal of the launcher / rendezvous / timing contract (gloo, a stand-in numpy step) — what the
world_size-2 CPU test runs; never a measurement.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0
# SURVEY.md §6 / BASELINE.md §2: the REFERENCE's own fp32 torch-CPU forward, measured in the survey container (8 vCPU Xeon
# 2.1 GHz, torch 2.10 CPU, 8 threads).  Carried beside the port's number so that the CPU baseline never flatters the ratio.
REFERENCE_TORCH_CPU_SURVEY = {"forward_latent_frames_per_s": [1408.0, 1548.0], "cores": 8,
                              "what": "reference JaT_AudioSR_V3 fp32 eval forward, B=2 / B=1, T=512, survey container"}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as a torch.distributed.run CHILD process (this parent
    never touches the GPU: no exec of an initialised process), stream their output through and return rank 0's JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               JAT_BENCH_CHILD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0 or line is None:
        raise SystemExit(f"bench ranks failed (exit {rc}); no result line")
    print(line)


def dry_run(args):
    """CPU rehearsal of the multi-rank contract (gloo): barrier + timed K stand-in steps + MAX over ranks + one JSON line
    from rank 0.  No GPU, no kernel: checks the launcher and the distributed plumbing only."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from jatsr_amd.dist import max_over_ranks, shard_range
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    a, b = shard_range(args.B, world, rank) if args.scaling == "strong" else (0, args.B)
    x = np.ones((b - a, 64), np.float32)
    metric, unit_per_step, scaling = "DiT latent-frames/sec (B=28,C=1024,T=512, 50-step CFG)", None, args.scaling
    if args.mode == "train":
        # the training step's only cross-rank traffic: the flat gradient buffer exchanged in slices (reduce-scatter + all-gather,
        # jatsr_amd.dist.exchange_sum_) — here on a CPU stand-in, through the same function and the same slice walk
        from jatsr_amd.dist import exchange_sum_
        grads = torch.full((3 * (1 << 19) + 6,), float(rank + 1))      # slices of 2 MiB: above the exchange's all_reduce cut-over
        metric, scaling = f"DiT training latent-frames/sec (B=28/GPU, C=1024, T={args.train_T})", "weak"

        def step():
            grads.fill_(float(rank + 1))
            n, nsl = grads.numel(), 3
            for sl in range(nsl):                              # slices as the backward hands them over
                lo, hi = sl * (n // nsl), n if sl == nsl - 1 else (sl + 1) * (n // nsl)
                for w in exchange_sum_(grads[lo:hi], async_op=True):
                    w.wait()
            assert float(grads[0]) == world * (world + 1) / 2 and float(grads[-1]) == world * (world + 1) / 2
            return float((x @ x.T).sum())
        unit_per_step = world * args.B * args.train_T
    elif args.mode == "long":
        # one long file: the chunk plan sharded round-robin over the ranks, one object gather per file (dist.sample_long_sharded)
        from jatsr_amd.dist import sample_long_sharded
        stride, chunk, ov = 1378 - 172, 1378, 172
        nchunk = (args.long_T - ov + stride - 1) // stride
        plan = [(i * stride, min(i * stride + chunk, args.long_T)) for i in range(nchunk)]
        metric, scaling = f"DiT long-sequence latent-frames/sec (one file, T={args.long_T}, 50-step CFG)", "strong"

        def step():
            chunks = sample_long_sharded(lambda idxs: {i: torch.full((1, 4, plan[i][1] - plan[i][0]), float(i)) for i in idxs}, plan)
            assert [c.shape[-1] for c in chunks] == [b_ - a_ for a_, b_ in plan] and all(float(c[0, 0, 0]) == i for i, c in enumerate(chunks))
            return float(sum(c.shape[-1] for c in chunks))
        unit_per_step = args.long_T
    else:
        def step():
            return float((x @ x.T).sum())
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t1)
    seen = world
    if world > 1:
        t = torch.tensor([1.0])
        dist.all_reduce(t)
        seen = int(t.item())
    total_b = args.B if args.scaling == "strong" else world * args.B
    per_step = unit_per_step if unit_per_step is not None else total_b * args.T
    if rank == 0:
        print(json.dumps({"metric": metric, "value": per_step * args.steps / elapsed,
                          "unit": "latent-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True, "ranks_seen": seen,
                          "config": {"workload": f"launcher rehearsal on CPU (gloo), stand-in step, mode {args.mode}", "B_local": b - a,
                                     "mode": args.mode}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--B", type=int, default=28)
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--num-steps", type=int, default=50, help="Euler steps per sampling run")
    ap.add_argument("--cfg-scale", type=float, default=3.0)
    ap.add_argument("--config", default="v3mod2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long", action="store_true", help="skip the T=4096 chunked-inference leg (configs[4])")
    ap.add_argument("--eager", action="store_true", help="replay the sampler without the hipGraph (A/B)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step leg (configs[3])")
    ap.add_argument("--mode", choices=["sample", "train", "long"], default="sample",
                    help="train: time K DDP training steps on every rank instead (configs[3]; not the headline metric); "
                         "long: one file of --long-T latent frames, its chunks sharded over the ranks (configs[4])")
    ap.add_argument("--long-T", type=int, default=4096)
    ap.add_argument("--train-T", type=int, default=1378)
    ap.add_argument("--latent-loss", type=float, default=0.3, help="--mode train: latent perceptual loss weight (0 = MSE)")
    ap.add_argument("--train-class", choices=["v3", "v2"], default="v3",
                    help="--mode train: JaT_AudioSR_V3 (RMSNorm; train_ddp_v3m2.py) or JaT_AudioSR_V2 (LayerNorm; the class "
                         "train_ddp_v3mod2.py:706 trains = BASELINE configs[3], with JAT_OPERAND_DTYPE=fp16 for its autocast dtype)")
    ap.add_argument("--no-single-chunk", action="store_true", help="skip the B=1 single-chunk sampler legs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: B per GPU fixed; strong: one batch of B sharded over the GPUs (dist.shard_range)")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the launcher and timing contract (tests)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])     # before anything touches a GPU
    if args.dry_run:
        return dry_run(args)

    if args.mode == "train":
        # the trainer overlaps the gradient exchange (its own stream) with the backward; with ROCm's default of four hardware
        # queues two streams of one process can land on the same queue and run one after the other (seen with two sampler
        # graphs on two streams: profiles/r03/two_stream_half_batches.log) — must be set before the HIP runtime starts
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch

    import jatsr_amd
    import jatsr_amd.recipe as recipe
    from jatsr_amd import _lib as L

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    L.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI; used only for the barrier + max-time

    cfg = recipe.CONFIGS[args.config]
    B, T, C_lat = args.B, args.T, cfg["input_channels"]
    from jatsr_amd.dist import shard_range
    if args.scaling == "strong" and args.mode == "sample":      # one batch of B, rank r takes rows [a, b)
        a_, b_ = shard_range(args.B, world, rank)
        B = b_ - a_
    ranks_seen = world
    if dist is not None:                                          # the ranks RCCL actually connected
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
    t0 = time.time()
    sd = recipe.make_state_dict(cfg)
    model = jatsr_amd.JaT_AudioSR_V3(**cfg, dropout=0.1, drop_path_rate=0.05)   # training-only rates, train_ddp_v3m2.py:82-83
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(dev).eval()
    lr = torch.from_numpy(recipe.gaussian("lr_latent", (B, C_lat, T), 1234 + 2 * rank)).to(dev)
    z0 = torch.from_numpy(recipe.gaussian("z0", (B, C_lat, T), 1235 + 2 * rank)).to(dev)
    sampler = jatsr_amd.Sampler(model, B, T, args.num_steps, args.cfg_scale)
    setup_s = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == "long":
        # configs[4]: ONE long file, reference chunk plan (1378-frame chunks, 172 overlap); rank r samples chunks r, r + W, ...
        # (dist.sample_long_sharded: no collective in the sampling loop, one object gather per file), rank 0 crossfades.
        # With one rank this is jatsr_amd.sample_long: every chunk of the file in ONE launch (short tail padded + masked).
        from jatsr_amd.dist import sample_long_sharded
        del sampler
        T_long = args.long_T
        lr_long = torch.from_numpy(recipe.gaussian("lr_long", (C_lat, T_long), 9)).to(dev)
        mean, std = torch.zeros(C_lat, device=dev), torch.ones(C_lat, device=dev)
        plan = jatsr_amd.chunk_plan(T_long)
        noise = [torch.from_numpy(recipe.gaussian("noise_long", (1, C_lat, b - a), i)).to(dev) for i, (a, b) in enumerate(plan)]

        def sample_chunks(idxs):
            """This rank's chunks: one launch per bucket of jatsr_amd.chunk_groups (normally ONE: rows shorter than the longest
            chunk padded + key-masked; a 128-token bucket cannot mask keys, so there every length gets its own launch)."""
            if not idxs:
                return {}
            all_lens = [plan[i][1] - plan[i][0] for i in idxs]
            out = {}
            for Tm, members in jatsr_amd.chunk_groups(all_lens).items():
                ids = [idxs[j] for j in members]
                lens = [all_lens[j] for j in members]
                lrb = torch.zeros(len(ids), C_lat, Tm, device=dev)
                zb = torch.zeros(len(ids), C_lat, Tm, device=dev)
                for j, i in enumerate(ids):
                    lrb[j, :, :lens[j]] = jatsr_amd.channel_affine(lr_long[None, :, plan[i][0]:plan[i][1]], mean, std)[0]
                    zb[j, :, :lens[j]] = noise[i][0]
                gen = jatsr_amd.flow_matching_sample(model, lrb, args.num_steps, args.cfg_scale, device=dev, verbose=False, z0=zb,
                                                     lengths=lens if any(v != Tm for v in lens) else None)
                gen = jatsr_amd.channel_affine(gen, mean, std, inverse=True)
                out.update({i: gen[j:j + 1, :, :lens[j]].contiguous() for j, i in enumerate(ids)})
            return out

        def one_file():
            chunks = sample_long_sharded(sample_chunks, plan)
            return jatsr_amd.crossfade_chunks([c.to(dev) for c in chunks], 172) if rank == 0 else None
        for _ in range(max(args.warmup, 1)):
            out = one_file()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out = one_file()
        barrier()
        elapsed = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        if rank == 0:
            assert out.shape == (1, C_lat, T_long) and bool(torch.isfinite(out).all())
            print(json.dumps({
                "metric": "DiT long-sequence latent-frames/sec (one file, T=%d, 50-step CFG)" % T_long,
                "value": T_long * args.steps / elapsed, "unit": "latent-frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": max(args.warmup, 1), "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": L.operand_dtype(), "data": "synthetic",
                "config": {"workload": f"{args.config} chunked inference of one file, T={T_long} -> chunks "
                                       f"{[b - a for a, b in plan]}, {args.num_steps}-step CFG={args.cfg_scale}, chunks "
                                       f"sharded round-robin over {world} rank(s), gathered once per file, crossfade on rank 0",
                           "parallelism": f"chunk-sharded x{world}", "rccl_ranks": ranks_seen}}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.mode == "train":
        # configs[3]: DDP training step, B per GPU, gradient all-reduce over RCCL overlapped with the backward.
        from jatsr_amd.train import Trainer
        del sampler
        Tt = args.train_T
        if args.train_class == "v2":      # the v3mod2 trainer's model class: LayerNorm without affine (train_ddp_v3mod2.py:706)
            del model
            torch.cuda.empty_cache()
            sd2 = recipe.make_state_dict(cfg, "ln")
            model = jatsr_amd.JaT_AudioSR_V2(**cfg, dropout=0.1, drop_path_rate=0.05)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()}, strict=False)
            model = model.to(dev).eval()
        # JAT_OPERAND_DTYPE=fp16 python bench.py --mode train: the v3mod2 trainer's fp16 autocast + dynamic loss scale
        trainer = Trainer(model, batch_size=B, frames=Tt, seed=1 + rank, latent_loss_weight=args.latent_loss)
        hr_t = torch.from_numpy(recipe.gaussian("train_hr", (B, C_lat, Tt), 300 + rank)).to(dev)
        lr_t = torch.from_numpy(recipe.gaussian("train_lr", (B, C_lat, Tt), 400 + rank)).to(dev)
        mean, std = torch.zeros(C_lat, device=dev), torch.ones(C_lat, device=dev)
        for _ in range(max(args.warmup, 1)):
            st = trainer.train_step(hr_t, lr_t, mean, std, mean, std)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            st = trainer.train_step(hr_t, lr_t, mean, std, mean, std)
        barrier()
        elapsed = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        assert np.isfinite(st["loss"]) and np.isfinite(st["grad_norm"])
        tfl = 3 * recipe.forward_flops(cfg, B, Tt) * args.steps / elapsed / 1e12
        if rank == 0:
            print(json.dumps({
                "metric": "DiT training latent-frames/sec (B=28/GPU, C=1024, T=%d)" % Tt,
                "value": world * B * Tt * args.steps / elapsed, "unit": "latent-frames/s", "n_gpus": world,
                "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": L.operand_dtype(), "data": "synthetic",
                "config": {"workload": f"{args.config} DDP training step, {'JaT_AudioSR_V2 (LayerNorm)' if args.train_class == 'v2' else 'JaT_AudioSR_V3 (RMSNorm)'}, "
                                       f"{L.operand_dtype()} operands (MSE + {args.latent_loss} x latent perceptual loss, "
                                       f"dropout 0.1, clip 1.0, AdamW), B={B}/GPU T={Tt}, flat-buffer gradient exchange "
                                       "over RCCL overlapped with the backward", "B_per_gpu": B, "T": Tt,
                           "parallelism": f"dp{world}", "model_class": "JaT_AudioSR_V2" if args.train_class == "v2" else "JaT_AudioSR_V3"},
                "tflops_per_gpu": tfl, "mfma_frac": tfl / PEAK_BF16_TFLOPS, "loss": st["loss"], "grad_norm": st["grad_norm"]}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    use_graph = not args.eager
    for _ in range(args.warmup):
        out = sampler.run(lr, z0, use_graph=use_graph)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = sampler.run(lr, z0, use_graph=use_graph)
    barrier()
    elapsed = time.perf_counter() - t1
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert bool(torch.isfinite(out).all()), "sampler produced non-finite values"
    total_B = args.B if args.scaling == "strong" else world * B
    frames = total_B * T * args.steps
    value = frames / elapsed

    fwd_flops_B = recipe.forward_flops(cfg, B, T)   # algorithmic FLOPs, SURVEY.md §8d closed form
    use_cfg = args.cfg_scale != 1.0
    run_flops = fwd_flops_B * (2 if use_cfg else 1) * args.num_steps
    sampler_tflops = run_flops * args.steps / elapsed / 1e12   # per GPU

    result = {
        "metric": "DiT latent-frames/sec (B=28,C=1024,T=512, 50-step CFG)",
        "value": value, "unit": "latent-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": L.operand_dtype(), "data": "synthetic",
        "config": {"workload": f"{args.config} DiT {args.num_steps}-step CFG={args.cfg_scale} flow-matching sampling, "
                               f"{'hipGraph' if use_graph else 'eager'}, "
                               + (f"B={B}/GPU" if args.scaling == "weak" else f"one batch of B={args.B} sharded "
                                  f"{[shard_range(args.B, world, r)[1] - shard_range(args.B, world, r)[0] for r in range(world)]}")
                               + f" C={C_lat} T={T}",
                   "B_per_gpu": B if args.scaling == "weak" else None, "global_batch": total_B, "T": T, "C": C_lat,
                   "euler_steps": args.num_steps, "cfg_scale": args.cfg_scale,
                   "weights": "recipe (random-init, non-zero adaLN/final)",
                   "parallelism": f"replicas x{world}, no collective in the loop", "rccl_ranks": ranks_seen},
        "sampler_mfma_frac": sampler_tflops / PEAK_BF16_TFLOPS,
        "sampler_tflops_per_gpu": sampler_tflops,
        "setup_s": setup_s,
    }

    if rank == 0:
        # ---- single forward (BASELINE configs[1]): B=28, T=512, per-sample t ---------------------------
        x_t = torch.from_numpy(recipe.gaussian("x_t", (B, C_lat, T), 77)).to(dev)
        tvec = torch.linspace(0.02, 0.98, B, device=dev)
        for _ in range(3):
            y = model(x_t, tvec, lr)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nf = 20
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(nf):
            y = model(x_t, tvec, lr)
        ev1.record()
        torch.cuda.synchronize()
        fwd_ms = ev0.elapsed_time(ev1) / nf
        result["forward"] = {"workload": f"single DiT forward B={B} T={T} (configs[1])", "ms": fwd_ms,
                             "latent_frames_per_s": B * T / (fwd_ms * 1e-3),
                             "tflops": fwd_flops_B / (fwd_ms * 1e-3) / 1e12,
                             "mfma_frac": fwd_flops_B / (fwd_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}

        # ---- the reference's own loop body: B = 1, one chunk per launch (infer_test_v3m2.py:370-398: a 16 s chunk = 1378 frames,
        # CFG double batch -> M = 690 rows), and one 512-frame chunk for comparison with the headline's T
        if not args.no_single_chunk and world == 1:
            legs = {}
            for Tc in (1378, 512):
                lr1 = torch.from_numpy(recipe.gaussian("lr_chunk", (1, C_lat, Tc), 21)).to(dev)
                z1 = torch.from_numpy(recipe.gaussian("z_chunk", (1, C_lat, Tc), 22)).to(dev)
                s1 = jatsr_amd.Sampler(model, 1, Tc, args.num_steps, args.cfg_scale)
                for _ in range(2):
                    o1 = s1.run(lr1, z1)
                torch.cuda.synchronize()
                tc0 = time.perf_counter()
                nrun = 5
                for _ in range(nrun):
                    o1 = s1.run(lr1, z1)
                torch.cuda.synchronize()
                tcs = (time.perf_counter() - tc0) / nrun
                assert bool(torch.isfinite(o1).all())
                fl1 = recipe.forward_flops(cfg, 1, Tc) * (2 if use_cfg else 1) * args.num_steps
                legs[f"T{Tc}"] = {"ms": tcs * 1e3, "latent_frames_per_s": Tc / tcs, "tflops": fl1 / tcs / 1e12,
                                  "mfma_frac": fl1 / tcs / 1e12 / PEAK_BF16_TFLOPS}
                del s1
            result["single_chunk"] = {"workload": f"B=1, one chunk per launch, {args.num_steps}-step CFG={args.cfg_scale}, hipGraph "
                                                  "(T1378: the reference's 16 s chunk, M = 690 rows with CFG)", **legs}

        # ---- long-sequence chunked inference (BASELINE configs[4]): one file of T=4096 latent frames -> 4 chunks
        # (3 x 1378 + 478, overlap 172; infer_test_v3m2.py:340-404), equal-length chunks batched, 50-step CFG each
        if not args.no_long and world == 1:   # single-GPU diagnostics: not repeated in the N > 1 scaling runs
            T_long = 4096
            lr_long = torch.from_numpy(recipe.gaussian("lr_long", (C_lat, T_long), 9)).to(dev)
            mean = torch.zeros(C_lat, device=dev)
            std = torch.ones(C_lat, device=dev)
            plan = jatsr_amd.chunk_plan(T_long)
            noise = [torch.from_numpy(recipe.gaussian("noise_long", (1, C_lat, b - a), i)).to(dev)
                     for i, (a, b) in enumerate(plan)]
            out_long = jatsr_amd.sample_long(model, lr_long, mean, std, mean, std, args.num_steps, args.cfg_scale, noise=noise)
            torch.cuda.synchronize()
            tl0 = time.perf_counter()
            out_long = jatsr_amd.sample_long(model, lr_long, mean, std, mean, std, args.num_steps, args.cfg_scale, noise=noise)
            torch.cuda.synchronize()
            tl = time.perf_counter() - tl0
            assert out_long.shape == (1, C_lat, T_long) and bool(torch.isfinite(out_long).all())
            result["long_sequence"] = {"workload": f"T={T_long} frames in {len(plan)} chunks {[b - a for a, b in plan]}, "
                                                   f"{args.num_steps}-step CFG={args.cfg_scale}, one GPU",
                                       "ms": tl * 1e3, "latent_frames_per_s": T_long / tl}

        # ---- roofline of the dominant kernel (MLP fc1 GEMM: 28 launches per forward, 26 % of the FLOPs), measured
        # live: the same sampling run replayed eagerly with every fc1 launch bracketed by a HIP event pair on the
        # launch stream (jat_prof_*); rocprofv3 --kernel-trace --stats of this command must agree (profiles/).
        per_run_launches = cfg["depth"] * args.num_steps
        L.check(L.lib().jat_prof_gemm_site(model._get_handle().ptr, 2, per_run_launches))
        sampler.run(lr, z0, use_graph=False)
        torch.cuda.synchronize()
        tot_ms, n_l, fl, var = C.c_double(), C.c_int32(), C.c_double(), C.c_int32()
        L.check(L.lib().jat_prof_collect(model._get_handle().ptr, C.byref(tot_ms), C.byref(n_l), C.byref(fl), C.byref(var)))
        Mg = (2 if use_cfg else 1) * B * ((T + 3) // 4)
        Ng, Kg = int(cfg["hidden_size"] * cfg.get("mlp_ratio", 4.0)), cfg["hidden_size"]
        g_ms = tot_ms.value / max(n_l.value, 1)
        g_flops = fl.value / max(n_l.value, 1)
        ach = g_flops / (g_ms * 1e-3) / 1e12
        traffic, traffic_src, traffic_build_ok = None, None, None   # fabric-side bytes per launch from the committed PMC passes
        import hashlib
        lib_sha = hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()[:16]
        for rnd in ("r03", "r02"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")))
            except Exception:
                continue
            ent = pmc.get(f"fc1_variant{var.value}_M{Mg}_N{Ng}_K{Kg}")
            if ent:
                traffic, traffic_src = ent.get("traffic_bytes"), f"profiles/{rnd}/pmc_traffic.json"
                traffic_build_ok = pmc.get("_lib_sha256_16") == lib_sha if pmc.get("_lib_sha256_16") else None
                break
        kname = "gemm_persist_kernel<2,4,7,5,EPI_BF16_GELU> (two 224x320 tiles per CU)" if var.value == 38 else "gemm_bf16_kernel<...,EPI_BF16_GELU>"
        result["roofline"] = {"kernel": f"{kname} tile variant {var.value} (MLP fc1) M={Mg} N={Ng} K={Kg}",
                              "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                              "traffic_note": "bytes/launch = (2*FETCH_SIZE+WRITE_SIZE)*1024 of THIS kernel variant and shape from "
                                              f"{traffic_src} (separate rocprofv3 --pmc passes, tools/pmc_traffic.sh; fabric-side: "
                                              "Infinity-Cache hits included); null if the variant/shape has no committed pass; "
                                              "traffic_same_build: the committed passes were taken with this very libjat_hip.so",
                              "traffic_same_build": traffic_build_ok, "lib_sha256_16": lib_sha,
                              "flops_per_launch": g_flops, "avg_launch_ms": g_ms, "launches_timed": n_l.value}

        # ---- training step (BASELINE configs[3]; train_ddp_v3m2.py:533-622): forward + MSE + backward + clip + AdamW on
        # hr, lr [B,1024,T] with U-shaped t, cond noise and CFG dropout; T=512 for comparability with the headline and
        # T=1378 (the reference's target_frames).  Runs last: it updates the weights.  FLOPs = 3 x forward closed form.
        if not args.no_train and world == 1:
            from jatsr_amd.train import Trainer
            del sampler
            torch.cuda.empty_cache()
            legs = {}
            mean, std = torch.zeros(C_lat, device=dev), torch.ones(C_lat, device=dev)
            for Tt, lw in ((T, 0.0), (1378, 0.0), (1378, 0.3)):   # lw = 0.3: + the v3mod2 latent perceptual (FFT) loss
                trainer = Trainer(model, batch_size=B, frames=Tt, seed=1, latent_loss_weight=lw, distributed=False)   # rank-0-only leg
                hr_t = torch.from_numpy(recipe.gaussian("train_hr", (B, C_lat, Tt), 300)).to(dev)
                lr_t = torch.from_numpy(recipe.gaussian("train_lr", (B, C_lat, Tt), 301)).to(dev)
                for _ in range(2):
                    st = trainer.train_step(hr_t, lr_t, mean, std, mean, std)
                torch.cuda.synchronize()
                ts0 = time.perf_counter()
                nt = 5
                for _ in range(nt):
                    st = trainer.train_step(hr_t, lr_t, mean, std, mean, std)
                torch.cuda.synchronize()
                ts = (time.perf_counter() - ts0) / nt
                assert np.isfinite(st["loss"]) and np.isfinite(st["grad_norm"])
                tfl = 3 * recipe.forward_flops(cfg, B, Tt) / ts / 1e12
                legs[f"T{Tt}" + ("_latent_loss" if lw else "")] = {"ms_per_step": ts * 1e3, "latent_frames_per_s": B * Tt / ts, "tflops": tfl,
                                  "mfma_frac": tfl / PEAK_BF16_TFLOPS, "loss": st["loss"], "grad_norm": st["grad_norm"],
                                  "workspace_GB": trainer.workspace_bytes() / 1e9}
                del trainer, hr_t, lr_t
                torch.cuda.empty_cache()
            # configs[3] as the reference runs it: JaT_AudioSR_V2 (LayerNorm) + MSE + latent perceptual loss under fp16 autocast with a
            # dynamic loss scale (train_ddp_v3mod2.py:706,745,854-896).  The operand dtype is a process-level choice (libjat_hip_fp16.so):
            # a CHILD process (this one keeps its GPU context; nothing is exec'ed) runs `bench.py --mode train --train-class v2`.
            import subprocess
            env = dict(os.environ, JAT_OPERAND_DTYPE="fp16")
            cmd = [sys.executable, os.path.abspath(__file__), "--mode", "train", "--train-class", "v2", "--train-T", "1378",
                   "--latent-loss", "0.3", "--steps", "5", "--warmup", "2", "--B", str(B)]
            try:
                cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
                if cp.returncode == 0 and line:
                    ch = json.loads(line[-1])
                    legs["T1378_v3mod2_V2_latent_loss_fp16"] = {"ms_per_step": ch["ms_per_step"], "latent_frames_per_s": ch["value"],
                                                               "tflops": ch["tflops_per_gpu"], "mfma_frac": ch["mfma_frac"],
                                                               "loss": ch["loss"], "grad_norm": ch["grad_norm"], "dtype": ch["dtype"],
                                                               "model_class": ch["config"]["model_class"]}
                else:
                    legs["T1378_v3mod2_V2_latent_loss_fp16"] = {"error": (cp.stderr or cp.stdout)[-400:]}
            except Exception as ex:   # the headline line must not die with a diagnostics leg
                legs["T1378_v3mod2_V2_latent_loss_fp16"] = {"error": repr(ex)[:400]}
            result["train_step"] = {"workload": f"{args.config} bf16 training step B={B}/GPU (fwd + MSE + bwd + "
                                                "clip_grad_norm 1.0 + AdamW), dropout 0.1 / DropPath 0..0.05 "
                                                "as train_ddp_v3m2.py:82-83 (JaT_AudioSR_V3, bf16 operands: the V3-class trainers); "
                                                "*_latent_loss: + 0.3 x latent perceptual loss; T1378_v3mod2_V2_latent_loss_fp16: "
                                                "BASELINE configs[3] as the reference runs it — JaT_AudioSR_V2 (LayerNorm), MSE + latent "
                                                "loss, fp16 operands + dynamic loss scale (train_ddp_v3mod2.py:706,745,854-896), child "
                                                "process on libjat_hip_fp16.so; one GPU; weight gradients and the re-pack's "
                                                "transposed copies run on the trainer's second stream (JAT_DW_STREAM=0: one stream)", **legs}

        # ---- CPU baseline: the numpy oracle (port of the reference's fp32 CPU forward, pinned to the
        # reference by tests/golden) on this host's cores: 1 warm-up + 3 timed forwards at B=28, T=512, and the oracle's
        # 50-step CFG sampler loop at B=2 time-boxed to its first steps (a full CPU run is ~20 min) and labelled as scaled.
        if world == 1 and not args.no_cpu_baseline:
            from oracle import jat_oracle as O
            from threadpoolctl import threadpool_limits
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            # BLAS thread count: BASELINE.md asks for the node's host cores; more threads are not faster for fp32 GEMMs of this size
            # (M = B * 128 rows), so a short sweep on a B = 4 forward picks the count and the sweep travels in the JSON.
            orc_s = O.OracleModel(cfg, sd, "rms", np.float32)
            xs4, xc4 = recipe.make_latents(4, C_lat, T, salt=4)
            t4 = np.linspace(0.1, 0.9, 4).astype(np.float32)
            sweep = {}
            for nthr in sorted({min(avail, n) for n in (16, 32, 64, 128, avail)}):
                threadpool_limits(limits=nthr)
                orc_s.forward(xs4, t4, xc4)
                w0 = time.perf_counter()
                orc_s.forward(xs4, t4, xc4)
                sweep[nthr] = time.perf_counter() - w0
            cores = min(sweep, key=sweep.get)
            del orc_s
            threadpool_limits(limits=cores)
            cpu_model = "unknown"
            try:
                for ln in open("/proc/cpuinfo"):
                    if ln.startswith("model name"):
                        cpu_model = ln.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            orc = O.OracleModel(cfg, sd, "rms", np.float32)
            xs2, xc2 = recipe.make_latents(2, C_lat, T, salt=5)
            orc.forward(xs2, np.array([0.3, 0.7], np.float32), xc2)            # warm-up (BLAS threads, page faults)
            Bc = args.B
            xs, xc = recipe.make_latents(Bc, C_lat, T, salt=6)
            tc = np.linspace(0.02, 0.98, Bc).astype(np.float32)
            reps = 3
            c0 = time.perf_counter()
            for _ in range(reps):
                orc.forward(xs, tc, xc)
            c_s = (time.perf_counter() - c0) / reps
            fwd_fps = Bc * T / c_s
            per_run = (2 if use_cfg else 1) * args.num_steps
            # the sampler loop itself (CFG double batch, combine, Euler update) on 2 samples, first steps only
            box_steps = 2
            lr2, z2 = recipe.gaussian("lr_latent", (2, C_lat, T), 1234), recipe.gaussian("z0", (2, C_lat, T), 1235)
            s0 = time.perf_counter()
            O.flow_matching_sample(orc, lr2, z2, args.num_steps, args.cfg_scale, max_steps=box_steps)
            s_s = (time.perf_counter() - s0) / box_steps                          # seconds per Euler step at B=2
            smp_fps = 2 * T / (s_s * args.num_steps)
            result["cpu_baseline"] = {
                "value": smp_fps, "unit": "latent-frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
                "host_cores_available": avail,
                "thread_sweep_s_per_B4_forward": {str(k): round(v, 3) for k, v in sweep.items()},
                "sample": f"oracle 50-step CFG={args.cfg_scale} sampler at B=2,T={T}: first {box_steps} of {args.num_steps} "
                          f"Euler steps timed ({s_s:.2f} s per step, CFG double batch) and scaled to the full run "
                          f"(time-boxed: a complete CPU run is ~{s_s * args.num_steps / 60 * 14:.0f} min at B=28); "
                          f"forward: 1 warm-up + {reps} timed fp32 numpy-oracle forwards at B={Bc},T={T} ({c_s:.2f} s each)",
                "forward_latent_frames_per_s": fwd_fps,
                "forward_gflops": fwd_flops_B / c_s / 1e9,
                "sampler_from_forward_rate": fwd_fps / per_run,
                "reference_torch_cpu_survey": REFERENCE_TORCH_CPU_SURVEY}
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
