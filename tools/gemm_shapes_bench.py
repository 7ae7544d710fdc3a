#!/usr/bin/env python3
"""GEMM tile-variant sweep on arbitrary (M, N, K, epilogue) shapes, e.g. the training step's at B=28, T=1378:

    python tools/gemm_shapes_bench.py --shapes 9660x1280x1280x1,9660x5120x1280x1,5120x1280x9728x0 --variants 18,20,21,25,26,27

epilogue: 0 = fp32 out (dW), 1 = bf16 out (forward / dX).  Variants are interleaved round-robin; median of `rounds`
samples of `reps` back-to-back launches (HIP events).  The auto column is what pick_variant chooses (variant -1)."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd._lib as L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", required=True)
    ap.add_argument("--variants", default="18,20,21,25,26,27,28,31,32,33,35")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    variants = [int(v) for v in a.variants.split(",")]
    for sh in a.shapes.split(","):
        M, N, K, epi = (int(x) for x in sh.split("x"))
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        out = torch.zeros(M, N, dtype=torch.float32 if epi == 0 else torch.bfloat16, device="cuda")
        samples = {v: [] for v in variants}
        ok = []
        for v in variants:
            rc = L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), None, L.ptr(out), M, N, K, epi, None, 0, M, v, L.stream_ptr())
            torch.cuda.synchronize()
            if rc == 0:
                ok.append(v)
        for _ in range(a.rounds):
            for v in ok:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), None, L.ptr(out), M, N, K, epi, None, 0, M, v, L.stream_ptr())
                e1.record()
                torch.cuda.synchronize()
                samples[v].append(e0.elapsed_time(e1) / a.reps * 1e3)
        fl = 2.0 * M * N * K
        line = "  ".join(f"v{v}: {statistics.median(samples[v]):6.1f}us {fl / statistics.median(samples[v]) / 1e6:5.0f}TF" for v in ok)
        best = min(ok, key=lambda v: statistics.median(samples[v]))
        print(f"M={M} N={N} K={K} epi={epi}: best v{best} | {line}", flush=True)


if __name__ == "__main__":
    main()
