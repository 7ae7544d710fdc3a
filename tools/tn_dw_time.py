"""Time the token-major weight-gradient GEMM (gemm_tn.hip) at the training step's shapes (B=28, T=1378 -> 9660 tokens)."""
import sys
import torch
sys.path.insert(0, ".")
import jatsr_amd._lib as L

L.require_gpu()
dev = torch.device("cuda:0")
OP = torch.float16 if L.OPERAND_DTYPE == "fp16" else torch.bfloat16
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 9660
for out, inn, ks in [(5120, 1280, 0), (1280, 5120, 0), (3840, 1280, 0), (1280, 1280, 0), (5120, 1280, 1), (5120, 1280, 3), (1280, 1280, 5)]:
    dY = (torch.randn(tokens, out, device=dev) * 0.05).to(OP)
    X = torch.randn(tokens, inn, device=dev).to(OP)
    dW = torch.empty(out, inn, device=dev)
    db = torch.empty(out, device=dev)
    work = torch.empty(64 + 16 * out * inn + 32 * out, device=dev)
    for with_db in (False, True):
        def run():
            L.check(L.lib().jat_k_weight_grad(L.ptr(dY), L.ptr(X), L.ptr(dW), L.ptr(db) if with_db else None, tokens, out, inn,
                                              ks, L.ptr(work), work.numel() * 4, L.stream_ptr()))
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"out {out:5d} in {inn:5d} ksplit {ks} db {int(with_db)}: {us:7.1f} us  {2.0 * tokens * out * inn / us / 1e6:7.1f} TF/s", flush=True)
