"""Which weight-gradient path is right after an optimiser step?  MSE-only loss: sum(final Linear bias grad) = scale * 2/N * sum(pred - target)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
import jatsr_amd
from jatsr_amd import recipe
from jatsr_amd.train import Trainer
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
B, Tt, C = 28, int(sys.argv[1]) if len(sys.argv) > 1 else 1378, cfg["input_channels"]
sd = recipe.make_state_dict(cfg)
hr = torch.from_numpy(recipe.gaussian("train_hr", (B, C, Tt), 300)).to(dev)
lr = torch.from_numpy(recipe.gaussian("train_lr", (B, C, Tt), 400)).to(dev)
LW = float(os.environ.get("AB_LW", "0.3"))
import jatsr_amd._lib as L
for flag in ("0", "1"):
    os.environ["JAT_TN_DW"] = flag
    model = jatsr_amd.JaT_AudioSR_V3(**cfg, dropout=0.1, drop_path_rate=0.05)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(dev).eval()
    tr = Trainer(model, batch_size=B, frames=Tt, seed=1, latent_loss_weight=LW, distributed=False, lr=5e-5)
    for step in range(3):
        z_t, t, cond = tr.prepare(hr, lr)
        pred = tr.forward_backward(z_t, t, cond, hr, want_pred=True, cond_clean=lr)
        torch.cuda.synchronize()
        off, n = [(o, c) for (k, o, c, *_r) in tr.layout if k == "final_layer.1.bias"][0]
        got = float(tr.grads[off:off + n].double().sum()) / tr.scaler.scale
        if LW == 0.0:
            want = float(2.0 * (pred.double() - hr.double()).sum() / pred.numel())
        else:
            ll = tr.latent_loss
            rows = B * C
            work = torch.empty((Tt * 8 + 255) // 256 * 256 + rows * 32, dtype=torch.uint8, device=dev)
            dref = torch.empty_like(pred); out6 = torch.zeros(6, device=dev)
            L.check(L.lib().jat_k_latent_loss(L.ptr(pred), L.ptr(hr), L.ptr(lr), L.ptr(dref), L.ptr(out6), rows, Tt,
                                              ll["latent_weight"], ll["freq_weight"], ll["ms_weight"], ll["consistency_weight"],
                                              ll["low_freq_phase_ratio"], ll["strict_cutoff"], ll["soft_cutoff"], 1.0,
                                              L.ptr(work), work.numel(), L.stream_ptr()))
            want = float(dref.double().sum())
            print("   loss terms", out6.tolist(), "|dpred_ref|", float(dref.double().norm()))
        gw = [(o, c) for (k, o, c, *_r) in tr.layout if k == "final_layer.1.weight"][0]
        print(f"TN={flag} step {step}: sum(db) {got:+.6e}  analytic {want:+.6e}   |db| {float(tr.grads[off:off+n].double().norm()):.4e} |dW| {float(tr.grads[gw[0]:gw[0]+gw[1]].double().norm()):.4e}", flush=True)
        tr.optimizer_step()
    del tr, model
