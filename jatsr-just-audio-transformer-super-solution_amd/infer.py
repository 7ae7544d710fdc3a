"""Inference driver — the MI355X counterpart of reference `infer_test_v3m2.py:main` (:236-450).

Same flags (`--checkpoint --val-dir --stats-file --output-dir --steps --cfg-scale --total-seconds --device
--input-file`), same chunk plan (16 s chunks = 1378 latent frames, 2 s = 172-frame linear crossfade, :340-404), same
per-channel normalisation (:381-394).  Differences, all outside the hot path's semantics:
  * equal-length chunks are batched into one captured sampler launch instead of the reference's serial B=1 loop;
  * the DAC decode + WAV writing of :411-437 is out of scope (the `descript-audio-codec` package and its weights are
    not available offline): the generated / HR / LR latents are written as a `.pt` file in the
    reference's latent container format, ready for `dac_codec.decode`;
  * `--seed` makes the initial noise reproducible (the reference draws it with torch.randn, :133).

    python -m jatsr_amd.infer --checkpoint ckpt.pt --input-file clip.pt --stats-file stats.json --cfg-scale 3.0
"""
from __future__ import annotations

import argparse
import os
import time

import torch

from . import io as jio
from .model import JaT_AudioSR_V2, JaT_AudioSR_V3, load_model
from .sampler import chunk_plan, sample_long


def build_parser():
    p = argparse.ArgumentParser(description="JaT-AudioSR V3 inference on MI355X (latent in, latent out)")
    p.add_argument("--checkpoint", type=str, default="checkpoints/v3_full_run/last.pt", help="V3 checkpoint path")
    p.add_argument("--val-dir", type=str, default="data_processed_v13_final/val", help="Validation latents directory")
    p.add_argument("--stats-file", type=str, default="data_processed_v13_final/global_stats_separated.json",
                   help="Normalization stats file (JSON or PT)")
    p.add_argument("--output-dir", type=str, default="inference_output_v3", help="Output directory")
    p.add_argument("--steps", type=int, default=50, help="Number of sampling steps")
    p.add_argument("--cfg-scale", type=float, default=1.0, help="CFG guidance scale (1.0 = no CFG)")
    p.add_argument("--total-seconds", type=float, default=None, help="Total output duration in seconds")
    p.add_argument("--device", type=str, default="cuda", help="Device (an AMD GPU; there is no CPU path)")
    p.add_argument("--input-file", type=str, default=None, help="Specific input file; default: first file in val-dir")
    p.add_argument("--layernorm", action="store_true", help="checkpoint is a v3mod2 (LayerNorm, JaT_AudioSR_V2) model")
    p.add_argument("--seed", type=int, default=None, help="seed for the initial noise")
    return p


def run(args):
    device = torch.device(args.device)
    os.makedirs(args.output_dir, exist_ok=True)
    model = load_model(args.checkpoint, device=device, cls=JaT_AudioSR_V2 if args.layernorm else JaT_AudioSR_V3)
    if args.input_file:
        path = args.input_file if os.path.exists(args.input_file) else os.path.join(args.val_dir, args.input_file)
        if not os.path.exists(path):
            raise FileNotFoundError(f"File not found: {path}")
    else:
        path = jio.first_latent_file(args.val_dir)
    hr, lr = jio.load_latent_file(path)
    C = model.input_channels
    stats = jio.load_stats(args.stats_file, channels=C, device=device)

    total = lr.shape[-1]
    if args.total_seconds is not None:
        total = min(total, jio.frames_for_seconds(args.total_seconds))
    chunk_frames, overlap = jio.frames_for_seconds(16.0), jio.frames_for_seconds(2.0)   # 1378, 172
    plan = chunk_plan(total, chunk_frames, overlap)
    print(f"input {os.path.basename(path)}: {total} frames -> {len(plan)} chunk(s) {[b - a for a, b in plan]}, "
          f"steps={args.steps}, cfg_scale={args.cfg_scale}")
    noise = None
    if args.seed is not None:
        g = torch.Generator(device="cpu").manual_seed(args.seed)
        noise = [torch.randn(1, C, b - a, generator=g).to(device) for a, b in plan]
    t0 = time.time()
    gen = sample_long(model, lr[:, :total].to(device), stats["hr_mean"], stats["hr_std"], stats["lr_mean"],
                      stats["lr_std"], num_steps=args.steps, cfg_scale=args.cfg_scale, chunk_frames=chunk_frames,
                      overlap_frames=overlap, noise=noise)
    torch.cuda.synchronize()
    dt = time.time() - t0
    stem = os.path.splitext(os.path.basename(path))[0]
    suffix = f"_cfg{args.cfg_scale:.1f}" if args.cfg_scale != 1.0 else ""
    out_path = os.path.join(args.output_dir, f"{stem}_generated{suffix}.pt")
    jio.save_latent_file(out_path, hr_latent=None if hr is None else hr[:, :total], lr_latent=lr[:, :total],
                         generated_latent=gen[0].to("cpu", torch.float16),
                         metadata={"source": os.path.basename(path), "steps": args.steps, "cfg_scale": args.cfg_scale,
                                   "frames": total, "seconds": dt})
    print(f"generated {gen.shape[-1]} frames in {dt:.2f} s -> {out_path}")
    return out_path


def main(argv=None):
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
