"""Small-M GEMM (one chunk: M = 256) with COLD weights: cycle through enough weight buffers that none survives in the Infinity
Cache, as in a sampling run (1 GB of weights per step).  us per launch by variant and K: fixed cost vs per-K-tile cost."""
import sys, torch
sys.path.insert(0, ".")
import jatsr_amd._lib as L
L.require_gpu()
dev = torch.device("cuda:0")
OP = torch.float16 if L.OPERAND_DTYPE == "fp16" else torch.bfloat16
M = 256
for N, Ks in [(5120, (320, 640, 1280, 2560)), (1792, (640, 1280, 2560)), (1280, (1280, 5120))]:
    for K in Ks:
        nbuf = max(4, int(600e6 / (N * K * 2)))
        Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(OP) for _ in range(nbuf)]
        A = torch.randn(M, K, device=dev).to(OP); bias = torch.zeros(N, device=dev); out = torch.zeros(M, N, dtype=OP, device=dev)
        res = []
        for v in (28, 20, 27):
            if N % {28: 128, 20: 128, 27: 160}[v]:
                continue
            def run(i):
                L.check(L.lib().jat_k_gemm(L.ptr(A), L.ptr(Ws[i % nbuf]), L.ptr(bias), L.ptr(out), M, N, K, 1, None, 0, 1, v, L.stream_ptr()))
            for i in range(nbuf): run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(3 * nbuf): run(i)
            e1.record(); torch.cuda.synchronize()
            res.append(f"v{v} {e0.elapsed_time(e1) / (3 * nbuf) * 1e3:6.1f}us")
        print(f"M {M} N {N} K {K} ({nbuf} weight buffers, {N * K * 2 / 1e6:.1f} MB each): " + "  ".join(res), flush=True)
