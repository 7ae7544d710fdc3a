#!/usr/bin/env python3
"""The bench's three in-process training legs (T=512, T=1378, T=1378 + latent loss) one after the other on one model."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jatsr_amd, jatsr_amd.recipe as recipe
from jatsr_amd.train import Trainer
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
model = jatsr_amd.JaT_AudioSR_V3(**cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg).items()}, strict=False)
model = model.to(dev)
B, C = 28, 1024
mean, std = torch.zeros(C, device=dev), torch.ones(C, device=dev)
legs = [(512, 0.0), (1378, 0.0), (1378, 0.3)] if len(sys.argv) < 2 else [(1378, 0.3)]
for Tt, lw in legs:
    trainer = Trainer(model, batch_size=B, frames=Tt, seed=1, latent_loss_weight=lw, distributed=False)
    hr = torch.from_numpy(recipe.gaussian("train_hr", (B, C, Tt), 300)).to(dev)
    lr = torch.from_numpy(recipe.gaussian("train_lr", (B, C, Tt), 301)).to(dev)
    for _ in range(2):
        st = trainer.train_step(hr, lr, mean, std, mean, std)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        st = trainer.train_step(hr, lr, mean, std, mean, std)
    torch.cuda.synchronize()
    print(f"T={Tt} lw={lw}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms  loss {st['loss']:.4f}  free {torch.cuda.mem_get_info()[0] / 1e9:.1f} GB")
