#!/usr/bin/env python3
"""Experiment: the B=28 CFG sampling run as TWO independent half-batch samplers (B=14 each: samples are independent, each half
keeps its own cond / uncond pairs) launched on two streams, so that the epilogue / prologue phases of one half overlap the K loops
of the other.  Prints the wall time of both halves together against the one B=28 sampler.   python tools/two_stream_ab.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd, jatsr_amd.recipe as recipe

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=28)
ap.add_argument("--T", type=int, default=512)
ap.add_argument("--runs", type=int, default=6)
ap.add_argument("--parts", type=int, default=2)
ap.add_argument("--eager", action="store_true")
ap.add_argument("--offset-us", type=float, default=0.0, help="delay stream i by i * this much before its launch (phase shift between the halves)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
model = jatsr_amd.JaT_AudioSR_V3(**cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg).items()}, strict=False)
model = model.to(dev).eval()
lr = torch.from_numpy(recipe.gaussian("lr_latent", (a.B, 1024, a.T), 1234)).to(dev)
z0 = torch.from_numpy(recipe.gaussian("z0", (a.B, 1024, a.T), 1235)).to(dev)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.runs):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


if a.parts == 1:
    whole = jatsr_amd.Sampler(model, a.B, a.T, 50, 3.0)
    out = []
    print(f"one sampler B={a.B}: {timed(lambda: out.append(whole.run(lr, z0))):.1f} ms  info={whole.info()}")
else:
    n = a.parts
    Bh = a.B // n
    parts = [jatsr_amd.Sampler(model, Bh, a.T, 50, 3.0) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    lrs = [lr[i * Bh:(i + 1) * Bh].contiguous() for i in range(n)]
    zs = [z0[i * Bh:(i + 1) * Bh].contiguous() for i in range(n)]
    outs = [None] * n

    def both():
        for i in range(n):
            with torch.cuda.stream(streams[i]):
                if a.offset_us > 0 and i > 0:
                    torch.cuda._sleep(int(a.offset_us * i * 100))       # the sleep kernel counts 100 MHz ticks on ROCm
                outs[i] = parts[i].run(lrs[i], zs[i], use_graph=not a.eager)
    ms = timed(both)
    print(f"{n} samplers B={Bh} on {n} streams (offset {a.offset_us} us): {ms:.1f} ms for all {a.B} samples  info={parts[0].info()} env={ {k: v for k, v in os.environ.items() if k.startswith('JAT_')} }")
    seq = timed(lambda: [parts[i].run(lrs[i], zs[i], use_graph=not a.eager) for i in range(n)])
    print(f"{n} samplers B={Bh} one after the other on one stream: {seq:.1f} ms")
