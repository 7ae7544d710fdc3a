"""Standalone timing of the attention forward at the training shape (B=28, N=345 tokens, 20 q heads / 4 kv heads)."""
import sys, torch
sys.path.insert(0, ".")
import jatsr_amd._lib as L
L.require_gpu()
dev = torch.device("cuda:0")
OP = torch.float16 if L.OPERAND_DTYPE == "fp16" else torch.bfloat16
for B, N in [(28, 345), (28, 128), (56, 128), (4, 1024)]:
    Hq, Hkv = 20, 4
    npad = (N + 63) // 64 * 64
    q = torch.randn(B * N, Hq * 64, device=dev).to(OP); k = torch.randn(B * N, Hkv * 64, device=dev).to(OP)
    vt = torch.zeros(B, Hkv, 64, npad, device=dev).to(OP); vt[..., :N] = torch.randn(B, Hkv, 64, N, device=dev).to(OP)
    o = torch.empty_like(q)
    def run():
        L.check(L.lib().jat_k_attention(L.ptr(q), L.ptr(k), L.ptr(vt), L.ptr(o), B, N, Hq, Hkv, npad, L.stream_ptr()))
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"B {B} N {N}: {us:7.1f} us  {4.0 * B * Hq * N * N * 64 / us / 1e6:7.1f} TF/s", flush=True)
