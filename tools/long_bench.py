#!/usr/bin/env python3
"""Long-sequence chunked inference (BASELINE configs[4]) on its own, for timing / rocprofv3:
    python tools/long_bench.py [--T 4096] [--steps 50]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    import torch
    import jatsr_amd, jatsr_amd.recipe as recipe
    cfg = recipe.CONFIGS["v3mod2"]
    C = cfg["input_channels"]
    model = jatsr_amd.JaT_AudioSR_V3(**cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg).items()}, strict=False)
    model = model.cuda().eval()
    lr = torch.from_numpy(recipe.gaussian("lr_long", (C, a.T), 9)).cuda()
    mean, std = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    plan = jatsr_amd.chunk_plan(a.T)
    noise = [torch.from_numpy(recipe.gaussian("noise_long", (1, C, e - s), i)).cuda() for i, (s, e) in enumerate(plan)]
    jatsr_amd.sample_long(model, lr, mean, std, mean, std, a.steps, 3.0, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        jatsr_amd.sample_long(model, lr, mean, std, mean, std, a.steps, 3.0, noise=noise)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(f"T={a.T} in {len(plan)} chunks {[e - s for s, e in plan]}: {dt * 1e3:.1f} ms, {a.T / dt:.0f} frames/s")


if __name__ == "__main__":
    main()
