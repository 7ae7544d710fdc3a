#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP training step vs a reference-autograd golden (tests/golden/train_*.npz), grouped
by layer and parameter kind: shows how the bf16 backward's rounding noise accumulates with depth.
    python tools/grad_err_table.py train_v3mod2_T128"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import load_golden, rel_l2
from test_gpu_train import make_trainer, step_inputs, gsub

name = sys.argv[1] if len(sys.argv) > 1 else "train_v3mod2_T128"
z, meta = load_golden(name)
m, tr = make_trainer(meta, use_grad_scaler=False, condition_noise_ratio=0.0)
hr, lr, noise, t, mask = step_inputs(meta)
z_t, t2, cond = tr.prepare(hr, lr, noise=noise, cfg_mask=mask, t=t)
tr.forward_backward(z_t, t2, cond, hr)
torch.cuda.synchronize()
print("loss", float(tr._scal[0]), "ref", float(z["loss64"]))
rows = collections.defaultdict(dict)
for k in meta["names"]:
    g = tr.grad(k).detach().cpu().numpy()
    r = rel_l2(gsub(g, meta), z["g_" + k])
    parts = k.split(".")
    if parts[0] == "blocks":
        rows[int(parts[1])][".".join(parts[2:])] = (r, float(z["gl2_" + k]))
    else:
        rows[-1][k] = (r, float(z["gl2_" + k]))
kinds = sorted({kk for L, d in rows.items() if L >= 0 for kk in d})
print("layer " + " ".join(f"{kk[-22:]:>22s}" for kk in kinds))
for L in sorted(k for k in rows if k >= 0):
    print(f"{L:5d} " + " ".join(f"{rows[L][kk][0]:10.2e}/{rows[L][kk][1]:9.2e}  " if kk in rows[L] else " " * 22 for kk in kinds))
for kk, (r, n) in rows[-1].items():
    print(f"{kk:40s} {r:10.2e} / {n:9.2e}")
