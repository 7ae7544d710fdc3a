#!/usr/bin/env python3
"""Does running independent batch slices as concurrent streams ("chains") fill the fixed-overhead gaps?
Splits a B=56 forward (the CFG double batch) into C slices, each on its own stream + workspace."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd, jatsr_amd.recipe as recipe
from jatsr_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=56)
ap.add_argument("--T", type=int, default=512)
ap.add_argument("--chains", default="1,2,4")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
sd = recipe.make_state_dict(cfg)
model = jatsr_amd.JaT_AudioSR_V3(**cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
model = model.to(dev).eval()
h = model._get_handle()
B, T = a.B, a.T
x_t = torch.randn(B, 1024, T, device=dev); x_c = torch.randn(B, 1024, T, device=dev)
t = torch.rand(B, device=dev); out = torch.empty_like(x_t)
ref = None
for nch in [int(c) for c in a.chains.split(",")]:
    per = B // nch
    streams = [torch.cuda.Stream() for _ in range(nch)]
    wss = []
    for _ in range(nch):
        need = C.c_size_t(); L.check(L.lib().jat_model_workspace_bytes(h.ptr, per, T, C.byref(need)))
        wss.append(torch.empty(need.value, dtype=torch.uint8, device=dev))
    def run():
        torch.cuda.current_stream().synchronize()
        for i, s in enumerate(streams):
            sl = slice(i * per, (i + 1) * per)
            L.check(L.lib().jat_forward(h.ptr, L.ptr(x_t[sl]), L.ptr(t[sl]), L.ptr(x_c[sl]), L.ptr(out[sl]), per, T,
                                        L.ptr(wss[i]), wss[i].numel(), C.c_void_p(s.cuda_stream)))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    if ref is None: ref = out.clone()
    print(f"chains={nch}: {dt*1e3:.3f} ms per B={B} forward; max diff vs 1 chain {float((out-ref).abs().max()):.2e}")
