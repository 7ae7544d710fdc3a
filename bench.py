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
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
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
    ap.add_argument("--mode", choices=["sample", "train"], default="sample",
                    help="train: time K DDP training steps on every rank instead (configs[3]; not the headline metric)")
    ap.add_argument("--train-T", type=int, default=1378)
    ap.add_argument("--latent-loss", type=float, default=0.3, help="--mode train: latent perceptual loss weight (0 = MSE)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import jatsr_amd
    import jatsr_amd.recipe as recipe
    from jatsr_amd import _lib as L

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    L.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI; used only for the barrier + max-time

    cfg = recipe.CONFIGS[args.config]
    B, T, C_lat = args.B, args.T, cfg["input_channels"]
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

    if args.mode == "train":
        # configs[3]: DDP training step, B per GPU, gradient all-reduce over RCCL overlapped with the backward.
        from jatsr_amd.train import Trainer
        del sampler
        Tt = args.train_T
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
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"{args.config} DDP training step (MSE + {args.latent_loss} x latent perceptual loss, "
                                       f"dropout 0.1, clip 1.0, AdamW), B={B}/GPU T={Tt}, flat-buffer gradient all-reduce "
                                       "over RCCL overlapped with the backward", "B_per_gpu": B, "T": Tt,
                           "parallelism": f"dp{world}"},
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
    frames = world * B * T * args.steps
    value = frames / elapsed

    fwd_flops_B = recipe.forward_flops(cfg, B, T)   # algorithmic FLOPs, SURVEY.md §8d closed form
    use_cfg = args.cfg_scale != 1.0
    run_flops = fwd_flops_B * (2 if use_cfg else 1) * args.num_steps
    sampler_tflops = run_flops * args.steps / elapsed / 1e12   # per GPU

    result = {
        "metric": "DiT latent-frames/sec (B=28,C=1024,T=512, 50-step CFG)",
        "value": value, "unit": "latent-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.config} DiT {args.num_steps}-step CFG={args.cfg_scale} flow-matching sampling, "
                               f"{'hipGraph' if use_graph else 'eager'}, B={B}/GPU C={C_lat} T={T}",
                   "B_per_gpu": B, "T": T, "C": C_lat, "euler_steps": args.num_steps, "cfg_scale": args.cfg_scale,
                   "weights": "recipe (random-init, non-zero adaLN/final)", "parallelism": f"replicas x{world}"},
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
        L.check(L.lib().jat_prof_gemm_site(2, per_run_launches))
        sampler.run(lr, z0, use_graph=False)
        torch.cuda.synchronize()
        tot_ms, n_l, fl, var = C.c_double(), C.c_int32(), C.c_double(), C.c_int32()
        L.check(L.lib().jat_prof_collect(C.byref(tot_ms), C.byref(n_l), C.byref(fl), C.byref(var)))
        Mg = (2 if use_cfg else 1) * B * ((T + 3) // 4)
        Ng, Kg = int(cfg["hidden_size"] * cfg.get("mlp_ratio", 4.0)), cfg["hidden_size"]
        g_ms = tot_ms.value / max(n_l.value, 1)
        g_flops = fl.value / max(n_l.value, 1)
        ach = g_flops / (g_ms * 1e-3) / 1e12
        traffic = None   # L2-miss bytes per launch from the committed PMC passes (same kernel, same shape), if any
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))
            traffic = pmc.get(f"fc1_variant{var.value}_M{Mg}_N{Ng}_K{Kg}", {}).get("traffic_bytes")
        except Exception:
            pass
        result["roofline"] = {"kernel": f"gemm_bf16_kernel<...,EPI_BF16_GELU> tile variant {var.value} (MLP fc1) "
                                        f"M={Mg} N={Ng} K={Kg}",
                              "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                              "traffic_note": "bytes/launch = (2*FETCH_SIZE+WRITE_SIZE)*1024 from profiles/r01/pmc_traffic.json "
                                              "(separate --pmc passes; fabric-side, Infinity-Cache hits included)",
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
            result["train_step"] = {"workload": f"{args.config} bf16 training step B={B}/GPU (fwd + MSE + bwd + "
                                                "clip_grad_norm 1.0 + AdamW), dropout 0.1 / DropPath 0..0.05 "
                                                "as train_ddp_v3m2.py:82-83; *_latent_loss: MSE + 0.3 x latent "
                                                "perceptual loss of train_ddp_v3mod2.py (configs[3]); one GPU", **legs}

        # ---- CPU baseline: numpy oracle (port of the reference fp32 CPU forward) on a bounded sample --------
        if world == 1 and not args.no_cpu_baseline:
            from oracle import jat_oracle as O
            from threadpoolctl import threadpool_limits
            cores = min(os.cpu_count() or 1, 32)     # BLAS threads actually used (more only oversubscribes)
            threadpool_limits(limits=cores)
            orc = O.OracleModel(cfg, sd, "rms", np.float32)
            Bc = 2
            xs, xc = recipe.make_latents(Bc, C_lat, T, salt=5)
            tc = np.array([0.3, 0.7], np.float32)[:Bc]
            orc.forward(xs, tc, xc)
            reps = 3
            c0 = time.perf_counter()
            for _ in range(reps):
                orc.forward(xs, tc, xc)
            c_s = (time.perf_counter() - c0) / reps
            fwd_fps = Bc * T / c_s
            per_run = (2 if use_cfg else 1) * args.num_steps
            result["cpu_baseline"] = {
                "value": fwd_fps / per_run, "unit": "latent-frames/s", "cores": cores, "kind": "port",
                "sample": f"{reps} timed fp32 numpy-oracle forwards at B={Bc},T={T} after 1 warm-up "
                          f"({c_s:.2f} s each, {fwd_fps:.0f} forward-frames/s); sampler rate = forward rate / {per_run} "
                          f"({args.num_steps} steps x CFG double batch) — scaled, the full CPU run (~20 min) is not executed",
                "forward_latent_frames_per_s": fwd_fps}
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
