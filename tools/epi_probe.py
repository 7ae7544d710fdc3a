#!/usr/bin/env python3
"""Epilogue anatomy of the sampler's GEMMs, stand-alone with random operands (run on the GPU box): interleaved rounds in one
process of {K loop only (JAT_GEMM_DBG=1), everything but the global stores (128), full} for the plain and the software-
pipelined epilogue variants, at a full-chip and a half-chip row count.

    python tools/epi_probe.py [--rounds 7]
"""
import argparse, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd._lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
OP = torch.float16 if L.OPERAND_DTYPE == "fp16" else torch.bfloat16
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def bench(cases, label):
    samples = {c[0]: [] for c in cases}
    for _ in range(a.rounds):
        for name, fn, dbg in cases:
            os.environ["JAT_GEMM_DBG"] = str(dbg)
            fn()
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            samples[name].append(e0.elapsed_time(e1) / a.reps * 1e3)
    os.environ["JAT_GEMM_DBG"] = "0"
    print(f"== {label}")
    for name, _, _ in cases:
        print(f"  {name:44s} {statistics.median(samples[name]):7.1f} us  (min {min(samples[name]):.1f})")


def fc1(M, N=5120, K=1280):
    A = torch.randn(M, K, device=dev).to(OP)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(OP)
    bias = torch.randn(N, device=dev) * 0.05
    part = torch.rand(M, 16, device=dev) * K / 16 + 0.1
    out = torch.zeros(M, N, dtype=OP, device=dev)

    def run(v, epi):
        return lambda: L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out), M, N, K, epi, None, 0, 128,
                                                       None, None, None, L.ptr(part), 16, v, L.stream_ptr()))
    cases = [("v31 K loop only", run(31, 2), 1)]
    for v in (31, 36, 38):
        cases += [(f"v{v} bf16 store, no GELU", run(v, 1), 0), (f"v{v} GELU, no global stores", run(v, 2), 128),
                  (f"v{v} GELU + stores (the sampler's fc1)", run(v, 2), 0)]
    bench(cases, f"fc1 consumer M={M} N={N} K={K} (rstd from 16 partials, bias)")


def resid(M, K, N=1280):
    A = torch.randn(M, K, device=dev).to(OP)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(OP)
    bias = torch.randn(N, device=dev) * 0.05
    gate = torch.randn(M // 128, N, device=dev) * 0.3
    x0 = torch.randn(M, N, device=dev)
    hi = x0.to(OP)
    lo = (x0 - hi.float()).to(OP)
    part = torch.zeros(M, 16, device=dev)

    def run(v):
        return lambda: L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), None, M, N, K, 3, L.ptr(gate), N, 128,
                                                       L.ptr(hi), L.ptr(lo), L.ptr(part), None, 0, v, L.stream_ptr()))
    cases = [("v32 K loop only", run(32), 1)]
    for v in (32, 39):
        cases += [(f"v{v} split residual, no global stores", run(v), 128), (f"v{v} split residual (the sampler's form)", run(v), 0)]
    bench(cases, f"gated residual producer M={M} N={N} K={K}")


for M in (7168, 3584):
    fc1(M)
resid(7168, 1280)
resid(7168, 5120)
resid(3584, 1280)
