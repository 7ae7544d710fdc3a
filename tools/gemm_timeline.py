#!/usr/bin/env python3
"""Where does a PIPE 6 slot spend its cycles?  s_memtime stamps around the four phases of the MFMA waves' loop
(fragment reads | barrier | 40 MFMAs | barrier), summed per wave (diagnostic path; shares, not absolute speed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd._lib as L

dev = torch.device("cuda:0")
M, N, K, variant, epi = 7168, 1280, 5120, 25, 3
A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device=dev) * 0.05; gate = torch.randn(M // 128, N, device=dev) * 0.3
out = torch.zeros(M, N, device=dev)
nblk = (M // 256) * (N // 160)
buf = torch.zeros(nblk * 8 * 4, dtype=torch.int64, device=dev)
os.environ["JAT_GEMM_TIMELINE"] = str(buf.data_ptr())
for _ in range(3):
    L.check(L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out), M, N, K, epi, L.ptr(gate), N, 128, variant, L.stream_ptr()))
torch.cuda.synchronize()
t = buf.view(nblk, 8, 4).double().cpu()
nk = K // 64
per = t.mean(dim=(0,)) / nk     # [wave][phase] cycles per K-tile
names = ["load(reads)", "barrier1", "mfma", "barrier2"]
for g, sl in (("group0 (waves 0-3)", slice(0, 4)), ("group1 (waves 4-7)", slice(4, 8))):
    v = per[sl].mean(0)
    print(g, {n: round(float(x), 1) for n, x in zip(names, v)}, "sum", round(float(v.sum()), 1), "cycles per K-tile")
