#!/usr/bin/env python3
"""Whole-kernel timeline of one GEMM launch from in-kernel stamps (the -DJAT_TIMELINE build: `make -C .../csrc libjat_hip_tl.so`,
run with JAT_LIB_PATH=.../libjat_hip_tl.so).  Per wave: s_memtime at entry, K loop start, K loop end, after the epilogue's first
barrier, exit; s_memrealtime (100 MHz) at entry and exit.  Prints medians over waves of each segment in cycles and ns, the clock
the chip held, and the launch's wall span (first entry to last exit).  A diagnostic build: read SHARES, not absolute speed.

    JAT_LIB_PATH=.../libjat_hip_tl.so python tools/tl_probe.py --shape fc1 --variant 36
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jatsr_amd._lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="fc1", choices=["fc1", "fc2", "out", "qkvattn"])
ap.add_argument("--variant", type=int, default=36)
ap.add_argument("--M", type=int, default=7168)
ap.add_argument("--dbg", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
OP = torch.bfloat16
M = a.M
N, K, epi = {"fc1": (5120, 1280, 2), "fc2": (1280, 5120, 3), "out": (1280, 1280, 3), "qkvattn": (1792, 1280, -1)}[a.shape]
A = torch.randn(M, K, device=dev).to(OP)
W = (torch.randn(N, K, device=dev) / K ** 0.5).to(OP)
bias = torch.randn(N, device=dev) * 0.05
gate = torch.randn(M // 128, N, device=dev) * 0.3
part_in = torch.rand(M, 16, device=dev) * K / 16 + 0.1
x0 = torch.randn(M, N, device=dev)
hi = x0.to(OP)
lo = (x0 - hi.float()).to(OP)
part_out = torch.zeros(M, 16, device=dev)
out = torch.zeros(M, N, dtype=OP, device=dev)
buf = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device=dev)
invf = torch.tensor([1.0 / 10000.0 ** (2 * i / 64.0) for i in range(32)], dtype=torch.float32, device=dev)
qbias = torch.randn(1792, device=dev) * 0.05
ao = torch.zeros(M, 1280, dtype=OP, device=dev)


def run():
    if epi == -1:
        L.check(L.lib().jat_k_qkv_attn(L.ptr(A), L.ptr(W), L.ptr(qbias), L.ptr(ao), M, 4, K, L.ptr(invf), L.ptr(part_in), 16,
                                       L.stream_ptr()))
    elif epi == 2:
        L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out), M, N, K, 2, None, 0, 128, None, None, None,
                                        L.ptr(part_in), 16, a.variant, L.stream_ptr()))
    else:
        L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), None, M, N, K, 3, L.ptr(gate), N, 128, L.ptr(hi),
                                        L.ptr(lo), L.ptr(part_out), None, 0, a.variant, L.stream_ptr()))


os.environ["JAT_GEMM_DBG"] = str(a.dbg)
for _ in range(20):          # warm clocks and caches without the stamps' stores
    run()
torch.cuda.synchronize()
os.environ["JAT_GEMM_TIMELINE"] = str(buf.data_ptr())
for _ in range(5):
    run()
torch.cuda.synchronize()
t = buf.view(-1, 16).cpu()
t = t[t[:, 0] != 0].double()
nb = int(t[:, 7].max()) + 1
seg = {"entry -> K loop start (prologue)": t[:, 1] - t[:, 0], "K loop": t[:, 2] - t[:, 1], "K loop end -> epilogue barrier passed": t[:, 3] - t[:, 2],
       "epilogue": t[:, 4] - t[:, 3], "whole wave": t[:, 4] - t[:, 0]}
real = (t[:, 6] - t[:, 5]) * 10.0       # ns
clk = float(((t[:, 4] - t[:, 0]) / real).median())   # cycles per ns = GHz
print(f"{a.shape} M={M} N={N} K={K} variant {a.variant} dbg={a.dbg}: {len(t)} waves of {nb} blocks, clock {clk:.2f} GHz")
for k, v in seg.items():
    print(f"  {k:42s} median {float(v.median()):9.0f} cyc = {float(v.median()) / clk / 1e3:6.2f} us   (p10 {float(v.quantile(0.1)) / clk / 1e3:6.2f}, p90 {float(v.quantile(0.9)) / clk / 1e3:6.2f})")
if epi == -1:
    x = t[:, 8:13]
    for k, v in {"RoPE + operand images (barrier -> images written)": x[:, 0] - t[:, 3], "wait at the image barrier": x[:, 1] - x[:, 0],
                 "Q fragments + 80 QK^T MFMAs": x[:, 2] - x[:, 1], "5 softmaxes (max, exp2, sum, pack)": x[:, 3] - x[:, 2],
                 "80 PV MFMAs": x[:, 4] - x[:, 3], "normalise + store": t[:, 4] - x[:, 4]}.items():
        print(f"    {k:52s} median {float(v.median()):9.0f} cyc = {float(v.median()) / clk / 1e3:6.2f} us   (p10 {float(v.quantile(0.1)) / clk / 1e3:6.2f}, p90 {float(v.quantile(0.9)) / clk / 1e3:6.2f})")
span = (float(t[:, 6].max()) - float(t[:, 5].min())) * 10.0 / 1e3
print(f"  launch span, first entry -> last exit (s_memrealtime) {span:7.2f} us; entries spread over {(float(t[:, 5].max()) - float(t[:, 5].min())) * 10 / 1e3:6.2f} us")
# second-round blocks: entries later than the earliest exit
first_exit = float(t[:, 6].min())
late = t[t[:, 5] >= first_exit]
print(f"  waves that entered after the first exit (second round): {len(late)}; their entry delay after the first exit: median {float(((late[:, 5] - first_exit) * 10).median()) / 1e3 if len(late) else 0:6.2f} us")
