#!/usr/bin/env python3
"""Run the same sampler / forward repeatedly and count bitwise mismatches (race detector)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jatsr_amd, jatsr_amd.recipe as recipe

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B, T, steps = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 64, 50)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
cfg = recipe.CONFIGS[cfg_name]
m = jatsr_amd.JaT_AudioSR_V3(**cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg).items()}, strict=False)
m = m.to("cuda").eval()
C = cfg["input_channels"]
lr = torch.from_numpy(recipe.gaussian("lr_latent", (B, C, T), 200)).cuda()
z0 = torch.from_numpy(recipe.gaussian("z0", (B, C, T), 201)).cuda()
ref = None; bad_g = bad_e = bad_f = 0
x_t = torch.cat([z0, z0]); tt = torch.full((2 * B,), 0.3, device="cuda"); xc = torch.cat([lr, torch.zeros_like(lr)])
fref = None
for i in range(iters):
    g = jatsr_amd.flow_matching_sample(m, lr, num_steps=steps, cfg_scale=3.0, verbose=False, z0=z0)
    e = jatsr_amd.flow_matching_sample(m, lr, num_steps=steps, cfg_scale=3.0, verbose=False, z0=z0, use_graph=False)
    f = m(x_t, tt, xc)
    if ref is None: ref, fref = g.clone(), f.clone()
    bad_g += int(not torch.equal(g, ref)); bad_e += int(not torch.equal(e, ref)); bad_f += int(not torch.equal(f, fref))
print(f"{cfg_name} B={B} T={T} steps={steps}: graph mismatches {bad_g}/{iters}, eager {bad_e}/{iters}, forward {bad_f}/{iters}"
      f"  env: GROUP={os.environ.get('JAT_ATTN_GROUP','1')} RPW={os.environ.get('JAT_NORM_RPW','4')} VAR={os.environ.get('JAT_GEMM_VARIANT','auto')}")
