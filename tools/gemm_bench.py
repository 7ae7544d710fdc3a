#!/usr/bin/env python3
"""GEMM variant microbenchmark on the block's four GEMM shapes (run on the GPU box).

    python tools/gemm_bench.py [--M 7168] [--variants 0,3,4,...] [--rounds 5]

Variants are interleaved round-robin in one process (cdna_hip_programming.md rule 24); each sample is `reps`
back-to-back launches timed with HIP events on the launch stream.  Prints median microseconds and TFLOP/s.
"""
import argparse
import statistics
import sys, os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd._lib as L

SHAPES = {"qkv": (1792, 1280, 1), "out": (1280, 1280, 3), "fc1": (5120, 1280, 2), "fc2": (1280, 5120, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=7168)
    ap.add_argument("--variants", default="18,21,25,31,32,33,35")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--shapes", default="qkv,out,fc1,fc2")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M = a.M
    variants = [int(v) for v in a.variants.split(",")]
    res = {}
    for name in a.shapes.split(","):
        N, K, epi = SHAPES[name]
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev) * 0.05
        gate = torch.randn(M // 128, N, device=dev) * 0.3
        out = torch.zeros(M, N, dtype=torch.float32 if epi == 3 else torch.bfloat16, device=dev)
        ref = (A.float() @ W.float().T + bias)

        def run(v):
            return L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out), M, N, K, epi, L.ptr(gate), N, 128, v,
                                      L.stream_ptr())
        ok = []
        for v in variants:
            out.zero_()
            rc = run(v)
            torch.cuda.synchronize()
            if rc != 0:
                continue
            if epi == 3:
                want = gate.repeat_interleave(128, 0) * ref
                err = float((out - want).norm() / want.norm())
            elif epi == 2:
                err = float((out.float() - torch.nn.functional.gelu(ref)).norm() / torch.nn.functional.gelu(ref).norm())
            else:
                err = float((out.float() - ref).norm() / ref.norm())
            if err > 5e-3:
                print(f"!! {name} variant {v}: rel err {err:.3e}")
            ok.append(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        samples = {v: [] for v in ok}
        for _ in range(a.rounds):
            for v in ok:
                run(v)
                e0.record()
                for _ in range(a.reps):
                    run(v)
                e1.record()
                torch.cuda.synchronize()
                samples[v].append(e0.elapsed_time(e1) / a.reps * 1e3)
        flops = 2.0 * M * N * K
        for v in ok:
            us = statistics.median(samples[v])
            res[(name, v)] = us
            print(f"{name:4s} M={M} N={N} K={K} v{v:<2d}: {us:8.1f} us  {flops / us / 1e6:7.0f} TFLOP/s   (min {min(samples[v]):.1f})")
    print("sum over block (qkv+out+fc1+fc2) per variant:")
    for v in variants:
        if all((n, v) in res for n in a.shapes.split(",")):
            print(f"  v{v}: {sum(res[(n, v)] for n in a.shapes.split(',')):.1f} us")
    best = {n: min(((res[(n, v)], v) for v in variants if (n, v) in res)) for n in a.shapes.split(",")}
    print("best per shape:", {n: (f"v{b[1]}", round(b[0], 1)) for n, b in best.items()}, "sum", round(sum(b[0] for b in best.values()), 1))


if __name__ == "__main__":
    main()
