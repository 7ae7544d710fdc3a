#!/usr/bin/env python3
"""Fixed-overhead probe: time one GEMM variant at several K (same M,N) and epilogues; the K->0 intercept is the
per-launch fixed cost (launch + prologue + epilogue), the slope the steady-state K-loop rate."""
import argparse, statistics, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd._lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=7168)
ap.add_argument("--N", type=int, default=5120)
ap.add_argument("--variants", default="31,32")
ap.add_argument("--Ks", default="64,320,1280,5120")
ap.add_argument("--epis", default="0,1,2,3")
a = ap.parse_args()
dev = torch.device("cuda:0")
M, N = a.M, a.N
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for v in [int(x) for x in a.variants.split(",")]:
    for epi in [int(x) for x in a.epis.split(",")]:
        line = []
        for K in [int(x) for x in a.Ks.split(",")]:
            A = torch.randn(M, K, device=dev).to(torch.bfloat16)
            W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device=dev) * 0.05
            gate = torch.randn(M // 128, N, device=dev) * 0.3
            out = torch.zeros(M, N, dtype=torch.float32 if epi in (0, 3) else torch.bfloat16, device=dev)
            def run():
                L.check(L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out), M, N, K, epi, L.ptr(gate), N, 128, v, L.stream_ptr()))
            ts = []
            for _ in range(5):
                run()
                e0.record()
                for _ in range(10):
                    run()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            line.append(f"K={K}: {statistics.median(ts):7.1f}us")
        print(f"v{v} N={N} epi{epi}  " + "  ".join(line))
