import sys, torch
sys.path.insert(0, ".")
import jatsr_amd._lib as L
L.require_gpu()
dev = torch.device("cuda:0")
tokens, out, inn, ks = 9660, 5120, 1280, int(sys.argv[1])
dY = (torch.randn(tokens, out, device=dev) * 0.05).to(torch.bfloat16); X = torch.randn(tokens, inn, device=dev).to(torch.bfloat16)
dW = torch.empty(out, inn, device=dev); work = torch.empty(64 + 16 * out * inn + 32 * out, device=dev)
def run():
    L.check(L.lib().jat_k_weight_grad(L.ptr(dY), L.ptr(X), L.ptr(dW), None, tokens, out, inn, ks, L.ptr(work), work.numel() * 4, L.stream_ptr()))
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"ks {ks}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
