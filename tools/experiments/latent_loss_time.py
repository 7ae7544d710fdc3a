"""Time the v3mod2 latent perceptual loss kernel (value + gradient) at the training shape: 28 x 1024 rows of T = 1378."""
import sys, torch
sys.path.insert(0, ".")
import jatsr_amd._lib as L
L.require_gpu()
dev = torch.device("cuda:0")
rows, T = 28 * 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1378
pred, hr, lr = (torch.randn(rows, T, device=dev) for _ in range(3))
work = torch.empty((T * 8 + 255) // 256 * 256 + rows * 32, dtype=torch.uint8, device=dev)
d = torch.empty_like(pred); out6 = torch.zeros(6, device=dev)
def run():
    L.check(L.lib().jat_k_latent_loss(L.ptr(pred), L.ptr(hr), L.ptr(lr), L.ptr(d), L.ptr(out6), rows, T, 0.3, 0.5, 0.5, 0.1, 0.3, 0.30, 0.36,
                                      1.0, L.ptr(work), work.numel(), L.stream_ptr()))
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"latent loss rows {rows} T {T}: {e0.elapsed_time(e1) / 10:.3f} ms  terms {out6.tolist()}")
