"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

import jatsr_amd.recipe as recipe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"])) if "meta" in z.files else {}
    return z, meta


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def sub(a, s1, s2):
    return a[:, ::s1, ::s2]


def fwd_inputs(meta):
    cfg = recipe.CONFIGS[meta["cfg"]]
    x_t, x_c = recipe.make_latents(meta["B"], cfg["input_channels"], meta["T"], salt=meta["salt"] + 100)
    t = np.asarray(meta["t"], dtype=np.float32)
    return cfg, x_t, t, x_c


def sampler_inputs(meta):
    cfg = recipe.CONFIGS[meta["cfg"]]
    C = cfg["input_channels"]
    lr = recipe.gaussian("lr_latent", (meta["B"], C, meta["T"]), meta["salt"] + 200)
    z0 = recipe.gaussian("z0", (meta["B"], C, meta["T"]), meta["salt"] + 201)
    return cfg, lr, z0
