#!/usr/bin/env python3
"""Time the stream-K gated-residual GEMM (variant 31) against the plain launch (25) on the sampler's two shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd._lib as L
print("CUs:", torch.cuda.get_device_properties(0).multi_processor_count, "G:", os.environ.get("JAT_STREAMK_G", "256"))
for (M, N, K) in ((7168, 1280, 5120), (7168, 1280, 1280)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    gate = torch.randn(M // 128, N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    for v in (25, 31, 25, 31):
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.check(L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), None, L.ptr(out), M, N, K, 3, L.ptr(gate), N, 128, v, L.stream_ptr()))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"M={M} N={N} K={K} variant {v}: {statistics.median(ts):.1f} us")
