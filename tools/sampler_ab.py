#!/usr/bin/env python3
"""Time the 50-step CFG sampler (B=28, T=512, hipGraph) under the current environment (JAT_GEMM_VARIANTS, JAT_* switches):
one line, no parity assertion — for A/B runs of the same box in consecutive processes.   python tools/sampler_ab.py [--B 28]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd, jatsr_amd.recipe as recipe

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=28)
ap.add_argument("--T", type=int, default=512)
ap.add_argument("--runs", type=int, default=8)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--tag", default="")
ap.add_argument("--zero", action="store_true", help="all-zero weights and inputs: same instructions, no operand toggling (power probe)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
sd = recipe.make_state_dict(cfg)
model = jatsr_amd.JaT_AudioSR_V3(**cfg)
model.load_state_dict({k: torch.from_numpy(v * 0 if a.zero else v) for k, v in sd.items()}, strict=False)
model = model.to(dev).eval()
lr = torch.from_numpy(recipe.gaussian("lr_latent", (a.B, 1024, a.T), 1234)).to(dev)
z0 = torch.from_numpy(recipe.gaussian("z0", (a.B, 1024, a.T), 1235)).to(dev)
if a.zero:
    lr, z0 = lr * 0, z0 * 0
sampler = jatsr_amd.Sampler(model, a.B, a.T, a.steps, 3.0)
out = None
for _ in range(a.warmup):
    out = sampler.run(lr, z0)
torch.cuda.synchronize()
ts = []
for _ in range(a.runs):
    t0 = time.perf_counter()
    out = sampler.run(lr, z0)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
env = {k: v for k, v in os.environ.items() if k.startswith("JAT_")}
print(f"sampler_ab {a.tag} B={a.B} T={a.T}: median {ts[len(ts) // 2]:.1f} ms  min {ts[0]:.1f} ms  finite={bool(torch.isfinite(out).all())} "
      f"checksum={float(out.double().abs().mean()):.6f} env={env}")
