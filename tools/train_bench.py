#!/usr/bin/env python3
"""Training-step timing / profiling target (the bench.py `train_step` leg on its own).

    python tools/train_bench.py --T 512 --steps 5
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -- python3 tools/train_bench.py --T 512 --steps 3
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="v3mod2")
    ap.add_argument("--B", type=int, default=28)
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--drop-path", type=float, default=0.05)
    ap.add_argument("--latent", type=float, default=0.0, help="latent perceptual loss weight (0.3 = the v3mod2 trainer)")
    ap.add_argument("--ln", action="store_true", help="JaT_AudioSR_V2 (LayerNorm), the model class of train_ddp_v3mod2.py")
    ap.add_argument("--split", action="store_true", help="time fwd+bwd and the optimiser separately")
    args = ap.parse_args()
    import numpy as np
    import torch

    import jatsr_amd
    import jatsr_amd.recipe as recipe
    from jatsr_amd.train import Trainer

    cfg = recipe.CONFIGS[args.config]
    C = cfg["input_channels"]
    cls = jatsr_amd.JaT_AudioSR_V2 if args.ln else jatsr_amd.JaT_AudioSR_V3
    model = cls(**cfg, dropout=args.dropout, drop_path_rate=args.drop_path)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg, "ln" if args.ln else "rms").items()},
                          strict=False)
    model = model.to("cuda")
    tr = Trainer(model, batch_size=args.B, frames=args.T, seed=1, latent_loss_weight=args.latent,
                 **(dict(cfg_dropout_prob=0.0, condition_noise_ratio=0.05) if args.latent else {}))
    hr = torch.from_numpy(recipe.gaussian("train_hr", (args.B, C, args.T), 300)).cuda()
    lr = torch.from_numpy(recipe.gaussian("train_lr", (args.B, C, args.T), 301)).cuda()
    mean, std = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    for _ in range(args.warmup):
        st = tr.train_step(hr, lr, mean, std, mean, std)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = tr.train_step(hr, lr, mean, std, mean, std)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    tfl = 3 * recipe.forward_flops(cfg, args.B, args.T) / (ms * 1e-3) / 1e12
    print(f"train step B={args.B} T={args.T}: {ms:.2f} ms, {args.B * args.T / ms * 1e3:.0f} frames/s, {tfl:.0f} TFLOP/s "
          f"(3x forward closed form), loss {st['loss']:.4f} gnorm {st['grad_norm']:.4f}, workspace {tr.workspace_bytes() / 1e9:.1f} GB")
    if args.split:
        z_t, t, cond = tr.prepare(hr, lr)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for _ in range(args.steps):
            tr.forward_backward(z_t, t, cond, hr)
        ev[1].record()
        for _ in range(args.steps):
            tr.optimizer_step()
        ev[2].record()
        torch.cuda.synchronize()
        print(f"  fwd+bwd {ev[0].elapsed_time(ev[1]) / args.steps:.2f} ms, optimiser+repack {ev[1].elapsed_time(ev[2]) / args.steps:.2f} ms")
    assert np.isfinite(st["loss"])


if __name__ == "__main__":
    main()
