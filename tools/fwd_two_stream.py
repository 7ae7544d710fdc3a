#!/usr/bin/env python3
"""Experiment: the B=28 single forward as two B=14 forwards on two streams (samples are independent) against one B=28 call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd, jatsr_amd.recipe as recipe
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
model = jatsr_amd.JaT_AudioSR_V3(**cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg).items()}, strict=False)
model = model.to(dev).eval()
B, T, n = 28, 512, 20
x = torch.from_numpy(recipe.gaussian("x_t", (B, 1024, T), 77)).to(dev)
lr = torch.from_numpy(recipe.gaussian("lr_latent", (B, 1024, T), 1234)).to(dev)
t = torch.linspace(0.02, 0.98, B, device=dev)
# a second module object = a second handle and workspace (one handle is not re-entrant)
model2 = jatsr_amd.JaT_AudioSR_V3(**cfg)
model2.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg).items()}, strict=False)
model2 = model2.to(dev).eval()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = B // 2
xa, xb, la, lb, ta, tb = x[:h].contiguous(), x[h:].contiguous(), lr[:h].contiguous(), lr[h:].contiguous(), t[:h].contiguous(), t[h:].contiguous()


def one():
    return model(x, t, lr)


def two():
    with torch.cuda.stream(s1):
        a = model(xa, ta, la)
    with torch.cuda.stream(s2):
        b = model2(xb, tb, lb)
    return a, b


for name, fn in (("one B=28 forward", one), ("two B=14 forwards on two streams", two)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per 28 samples")
