"""A/B of the weight-gradient paths at the benchmarked training shape: one train_step's flat gradient with JAT_TN_DW=0 / 1."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
import jatsr_amd
from jatsr_amd import recipe
from jatsr_amd.train import Trainer
dev = torch.device("cuda:0")
cfg = recipe.CONFIGS["v3mod2"]
B, Tt, C = int(sys.argv[1]) if len(sys.argv) > 1 else 28, int(sys.argv[2]) if len(sys.argv) > 2 else 1378, cfg["input_channels"]
sd = recipe.make_state_dict(cfg)
hr = torch.from_numpy(recipe.gaussian("train_hr", (B, C, Tt), 300)).to(dev)
lr = torch.from_numpy(recipe.gaussian("train_lr", (B, C, Tt), 400)).to(dev)
mean, std = torch.zeros(C, device=dev), torch.ones(C, device=dev)
res = {}
for flag in ("0", "1"):
    os.environ["JAT_TN_DW"] = flag
    model = jatsr_amd.JaT_AudioSR_V3(**cfg, dropout=0.1, drop_path_rate=0.05)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(dev).eval()
    tr = Trainer(model, batch_size=B, frames=Tt, seed=1, latent_loss_weight=0.3, distributed=False, lr=float(os.environ.get("AB_LR", "0")))
    for _ in range(int(os.environ.get("AB_STEPS", "1"))):
        st = tr.train_step(hr, lr, mean, std, mean, std)
        print("  step", st["loss"], st["grad_norm"], flush=True)
        for (n, off, cnt, *_r) in tr.layout:
            if n.startswith("blocks.26.attn.") and n.endswith("weight"):
                print("     ", n, float(tr.grads[off:off + cnt].double().norm()), float(tr.params[off:off + cnt].double().norm()))
    torch.cuda.synchronize()
    res["p" + flag] = tr.params.clone()
    res[flag] = (tr.grads.clone(), st, [(n, off, cnt) for (n, off, cnt, *_r) in tr.layout] if hasattr(tr, "layout") else None)
    print(flag, st["loss"], st["grad_norm"], flush=True)
    del tr, model
g0, g1 = res["0"][0].double(), res["1"][0].double()
print("param rel diff", float((res["p0"].double() - res["p1"].double()).norm() / res["p0"].double().norm()), "max abs", float((res["p0"] - res["p1"]).abs().max()), "n differing", int((res["p0"] != res["p1"]).sum()))
print("flat rel diff", float((g0 - g1).norm() / g0.norm()))
lay = res["0"][2]
if lay:
    rows = []
    for n, off, cnt in lay:
        a, b = g0[off:off + cnt], g1[off:off + cnt]
        rows.append((float((a - b).norm() / a.norm().clamp_min(1e-30)), n, float(a.norm()), float(b.norm())))
    print("--- backward order (last layers first)")
    for r in [x for x in rows if not x[1].startswith("blocks.")] + [x for x in rows if x[1].startswith("blocks.27.") or x[1].startswith("blocks.26.")]:
        print("%.3e  %-50s |old| %.4e |tn| %.4e" % r)
    rows.sort(reverse=True)
    for r in rows[:25]:
        print("%.3e  %-50s |old| %.4e |tn| %.4e" % r)
