#!/usr/bin/env python3
"""Board power and clocks (rocm-smi) sampled while the bench's sampler runs back to back — is the run pinned at the power cap?
    python tools/power_probe.py [--seconds 12]"""
import argparse, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd, jatsr_amd.recipe as recipe

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=12.0)
ap.add_argument("--train", action="store_true", help="load = the training step (B=28, T=1378) instead of the sampler")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
model = jatsr_amd.JaT_AudioSR_V3(**cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg).items()}, strict=False)
model = model.to(dev).eval()
lr = torch.from_numpy(recipe.gaussian("lr_latent", (28, 1024, 512), 1234)).to(dev)
z0 = torch.from_numpy(recipe.gaussian("z0", (28, 1024, 512), 1235)).to(dev)
if a.train:
    from jatsr_amd.train import Trainer
    model.train()
    trainer = Trainer(model, batch_size=28, frames=1378, seed=1, distributed=False)
    hr_t = torch.from_numpy(recipe.gaussian("train_hr", (28, 1024, 1378), 300)).to(dev)
    lr_t = torch.from_numpy(recipe.gaussian("train_lr", (28, 1024, 1378), 301)).to(dev)
    mean, std = torch.zeros(1024, device=dev), torch.ones(1024, device=dev)
    step = lambda: trainer.train_step(hr_t, lr_t, mean, std, mean, std)
else:
    sampler = jatsr_amd.Sampler(model, 28, 512, 50, 3.0)
    step = lambda: sampler.run(lr, z0)
step()
torch.cuda.synchronize()


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return f"rocm-smi failed: {e}"
    keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("Power", "sclk", "mclk", "fclk", "Temperature (Sensor junction)", "Max Graphics"))]
    return " | ".join(keep[:10])


print("idle :", smi())
stop = False


def load():
    while not stop:
        step()
        torch.cuda.synchronize()


th = threading.Thread(target=load)
th.start()
t0 = time.time()
while time.time() - t0 < a.seconds:
    time.sleep(2.0)
    print(f"t={time.time() - t0:5.1f}s busy:", smi())
stop = True
th.join()
