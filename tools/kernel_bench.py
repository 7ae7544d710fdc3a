#!/usr/bin/env python3
"""Micro-benchmark of the non-GEMM kernels at the sampler's shapes (B=56 CFG batch, N=128, D=1280)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import jatsr_amd._lib as L

dev = torch.device("cuda:0")
B, N, D, Hq, Hkv = 56, 128, 1280, 20, 4
M = B * N
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def timeit(fn, reps=20, rounds=5):
    ts = []
    for _ in range(rounds):
        fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)

x = torch.randn(M, D, device=dev); w = torch.rand(D, device=dev) + 0.5
mod = torch.randn(1, 2 * D, device=dev) * 0.3
y = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
def norm():
    L.check(L.lib().jat_k_norm_modulate(L.ptr(x), L.ptr(w), C.c_void_p(mod.data_ptr()), C.c_void_p(mod.data_ptr() + 4 * D), 0,
                                        L.ptr(y), M, D, N, 0, L.stream_ptr()))
us = timeit(norm)
print(f"norm_modulate M={M} D={D}: {us:.2f} us  {(M*D*6)/us/1e6:.2f} TB/s  (JAT_NORM_RPW={os.environ.get('JAT_NORM_RPW','4')})")

q = torch.randn(M, Hq * 64, device=dev).to(torch.bfloat16); k = torch.randn(M, Hkv * 64, device=dev).to(torch.bfloat16)
vt = torch.randn(B, Hkv, 64, N, device=dev).to(torch.bfloat16); o = torch.empty_like(q)
def attn():
    L.check(L.lib().jat_k_attention(L.ptr(q), L.ptr(k), L.ptr(vt), L.ptr(o), B, N, Hq, Hkv, N, L.stream_ptr()))
us = timeit(attn)
print(f"attention B={B} N={N}: {us:.2f} us  {4*B*Hq*N*N*64/us/1e6:.1f} TFLOP/s  (JAT_ATTN_KVB={os.environ.get('JAT_ATTN_KVB','128')})")
