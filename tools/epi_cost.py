#!/usr/bin/env python3
"""Where does a GEMM launch spend its time outside the K loop?  One process, interleaved rounds (cdna guide rule 24):
   (a) JAT_GEMM_DBG=1: prologue + K loop only (the coalesced epilogue returns right after its first barrier),
   (b) the epilogue under test.   python tools/epi_cost.py [--variant 31 --N 5120 --K 1280 --epis 1,2]"""
import argparse, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd._lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=7168)
ap.add_argument("--N", type=int, default=5120)
ap.add_argument("--K", type=int, default=1280)
ap.add_argument("--variant", type=int, default=31)
ap.add_argument("--epis", default="1,2")
a = ap.parse_args()
dev = torch.device("cuda:0")
M, N, K, v = a.M, a.N, a.K, a.variant
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device=dev) * 0.05
gate = torch.randn(M // 128, N, device=dev) * 0.3
outs = {e: torch.zeros(M, N, dtype=torch.float32 if e in (0, 3) else torch.bfloat16, device=dev) for e in (0, 1, 2, 3)}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def run(epi, dbg):
    os.environ["JAT_GEMM_DBG"] = str(dbg)
    L.check(L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(outs[epi]), M, N, K, epi, L.ptr(gate), N, 128, v, L.stream_ptr()))


cases = [("K loop only (dbg=1)", int(a.epis.split(",")[0]), 1)] + [(f"epilogue {e}", int(e), 0) for e in a.epis.split(",")]
samples = {c[0]: [] for c in cases}
for _ in range(7):
    for name, epi, dbg in cases:
        run(epi, dbg)
        e0.record()
        for _ in range(10):
            run(epi, dbg)
        e1.record()
        torch.cuda.synchronize()
        samples[name].append(e0.elapsed_time(e1) / 10 * 1e3)
base = None
for name, _, _ in cases:
    us = statistics.median(samples[name])
    base = us if base is None else base
    print(f"v{v} M={M} N={N} K={K}  {name:22s} {us:7.1f} us   (+{us - base:5.1f} us over the K loop)   {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s")
