#!/usr/bin/env python3
"""The single DiT forward of BASELINE configs[1] (B=28, T=512, per-sample t) N times — the workload of bench.py's `forward`
leg, alone, for rocprofv3 --kernel-trace (tools/prof_agg.py aggregates).   python tools/fwd_prof.py [--n 20] [--B 28]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd, jatsr_amd.recipe as recipe

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20)
ap.add_argument("--B", type=int, default=28)
ap.add_argument("--T", type=int, default=512)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
sd = recipe.make_state_dict(cfg)
model = jatsr_amd.JaT_AudioSR_V3(**cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
model = model.to(dev).eval()
x_t = torch.from_numpy(recipe.gaussian("x_t", (a.B, 1024, a.T), 77)).to(dev)
lr = torch.from_numpy(recipe.gaussian("lr_latent", (a.B, 1024, a.T), 1234)).to(dev)
tvec = torch.linspace(0.02, 0.98, a.B, device=dev)
for _ in range(3):
    y = model(x_t, tvec, lr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.n):
    y = model(x_t, tvec, lr)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.n * 1e3
fl = recipe.forward_flops(cfg, a.B, a.T)
print(f"forward B={a.B} T={a.T}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s  {fl / ms / 1e9 / 2500 * 100:.1f} % of the bf16 MFMA peak  finite={bool(torch.isfinite(y).all())}")
