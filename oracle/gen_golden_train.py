#!/usr/bin/env python3
"""Generate tests/golden/train_*.npz: one optimisation step of the REFERENCE model under torch autograd.

Build container only (needs /root/reference and CPU PyTorch):

    python oracle/gen_golden_train.py            # all cases
    python oracle/gen_golden_train.py micro      # a subset

What is pinned (training row a14 of SURVEY.md §8; the step body lives inline in `main()` of
train_ddp_v3m2.py:533-622, which cannot be imported — it needs tensorboard and a process group — so the few tensor
statements of the step are issued here against the reference's own model class, `torch.nn.functional.mse_loss`,
`torch.nn.utils.clip_grad_norm_` and `torch.optim.AdamW`, i.e. the very functions the trainer calls):

    z_t   = t * hr_norm + (1 - t) * noise                    train_ddp_v3m2.py:577-579
    lr_in = lr_norm * (~cfg_mask)                            :568-571
    pred  = model(z_t, t, lr_in)                             :582     (train mode, dropout = drop_path = 0)
    loss  = mse_loss(pred, hr_norm)                          :585
    loss.backward(); clip_grad_norm_(params, 1.0)            :610-615
    AdamW(lr, weight_decay=0.1).step()                       :423,618

`u_shaped_timestep_sampling` (:164-172) is a top-level function and is taken from the file itself with `ast`.
Stochastic regularisers (Dropout / DropPath, torch's Philox streams) are outside what a fixture can pin; the cases
run with both rates 0.  Only inputs-by-recipe metadata and expected VALUES are written (subsampled for the large
matrices, plus their full L2 norms).
"""
from __future__ import annotations

import ast
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import jatsr_amd.recipe as recipe  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FULL_LIMIT = 4096          # tensors up to this many elements are stored whole


def ref_model(cfg, norm, salt, dtype):
    with contextlib.redirect_stdout(io.StringIO()):
        if norm == "rms":
            from src.models.jat_audiosr_v3 import JaT_AudioSR_V3 as Cls
        else:
            from src.models.jat_audiosr_v2 import JaT_AudioSR_V2 as Cls
        m = Cls(**cfg, dropout=0.0, drop_path_rate=0.0)
    sd = {k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg, norm, salt).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(".rope." in k for k in missing)
    return m.to(dtype).train()


def sub(a, strides, stride1d=None):
    """Values are stored as fp32 (of the fp64 run): the HIP path's own error is 1e-3..1e-2.
    stride1d: sub-sampling step of non-matrix tensors above FULL_LIMIT (default: the product of the matrix strides)."""
    a = np.asarray(a, dtype=np.float32)
    if a.size <= FULL_LIMIT or a.ndim != 2:
        return np.ascontiguousarray(a if a.size <= FULL_LIMIT else a.reshape(-1)[::(stride1d or strides[0] * strides[1])])
    return np.ascontiguousarray(a[::strides[0], ::strides[1]])


def step_inputs(cfg, B, T, salt):
    C = cfg["input_channels"]
    hr = recipe.gaussian("train_hr", (B, C, T), salt + 300)
    lr = recipe.gaussian("train_lr", (B, C, T), salt + 301)
    noise = recipe.gaussian("train_noise", (B, C, T), salt + 302)
    return hr, lr, noise


def ref_charbonnier():
    """`charbonnier_loss` of train_ddp_v3m2mod1.py:72-101, taken from the file itself with `ast` (the module cannot be
    imported: tensorboard, process group)."""
    src = open(os.path.join(REF, "train_ddp_v3m2mod1.py"), encoding="utf-8").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "charbonnier_loss"]
    assert len(fn) == 1
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "train_ddp_v3m2mod1.py", "exec"), ns)
    return ns["charbonnier_loss"]


def charbonnier_case(name, B, C, T, salt=0, eps=1e-6):
    """Value and d/d pred of the reference's charbonnier_loss (train_ddp_v3m2mod1.py:72-101,666-672) on recipe tensors; a third
    of the elements have |pred - target| around sqrt(eps) or exactly 0, where the loss leaves its L1 regime."""
    fn = ref_charbonnier()
    pred_np = recipe.gaussian("charb_pred", (B, C, T), salt + 500)
    target_np = recipe.gaussian("charb_target", (B, C, T), salt + 501)
    near = recipe.gaussian("charb_near", (B, C, T), salt + 502)
    flat_t, flat_p, flat_n = target_np.reshape(-1), pred_np.reshape(-1), near.reshape(-1)
    flat_t[::3] = flat_p[::3] + 2e-3 * flat_n[::3]       # |d| ~ sqrt(eps) = 1e-3
    flat_t[::9] = flat_p[::9]                            # d == 0: gradient exactly 0, loss sqrt(eps)
    rec = {}
    for tag, dt in (("64", torch.float64), ("32", torch.float32)):
        pred = torch.from_numpy(pred_np).to(dt).requires_grad_(True)
        loss = fn(pred, torch.from_numpy(target_np).to(dt), eps=eps)
        loss.backward()
        rec["loss" + tag] = np.float64(loss.item())
        rec["dpred" + tag] = pred.grad.numpy()
    rec["meta"] = json.dumps(dict(case=name, B=B, C=C, T=T, salt=salt, eps=eps, torch=torch.__version__))
    np.savez_compressed(os.path.join(GOLD, f"train_charbonnier_{name}.npz"), **rec)
    print(f"[golden] train_charbonnier_{name}: loss64={rec['loss64']:.8f} loss32-loss64={rec['loss32'] - rec['loss64']:.2e}")


def train_case(name, cfg_name, B, T, t_list, mask_list, norm="rms", salt=0, lr_rate=5e-5, wd=0.1, clip=1.0,
               strides=(7, 5), loss_name="mse", charbonnier_eps=1e-6):
    """loss_name: "mse" (train_ddp_v3m2.py:585) or "charbonnier" (the V3M2-MOD1 trainer, train_ddp_v3m2mod1.py:666-672)."""
    charb = ref_charbonnier() if loss_name == "charbonnier" else None
    cfg = recipe.CONFIGS[cfg_name]
    hr, lr, noise = step_inputs(cfg, B, T, salt)
    t = np.asarray(t_list, dtype=np.float32)
    mask = np.asarray(mask_list, dtype=bool)
    rec = {}
    for tag, dt in (("64", torch.float64), ("32", torch.float32)):
        m = ref_model(cfg, norm, salt, dt)
        hr_t, lr_t, nz = (torch.from_numpy(a).to(dt) for a in (hr, lr, noise))
        tt = torch.from_numpy(t).to(dt)
        cfg_mask = torch.from_numpy(mask).view(B, 1, 1)
        lr_in = lr_t * (~cfg_mask).to(dt)
        tv = tt.view(-1, 1, 1)
        z_t = tv * hr_t + (1 - tv) * nz
        opt = torch.optim.AdamW(m.parameters(), lr=lr_rate, weight_decay=wd)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        pred = m(z_t, tt, lr_in)
        loss = charb(pred, hr_t, eps=charbonnier_eps) if charb else torch.nn.functional.mse_loss(pred, hr_t)
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        gnorm = torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
        opt.step()
        rec[f"loss{tag}"] = np.float64(loss.item())
        rec[f"gnorm{tag}"] = np.float64(gnorm.item())
        if tag == "64":
            rec["pred_l2"] = np.float64(pred.detach().norm().item())
            for k, g in grads.items():
                rec["g_" + k] = sub(g.numpy(), strides)
                rec["gl2_" + k] = np.float64(g.norm().item())
                d = (dict(m.named_parameters())[k].detach() - before[k]).numpy()
                rec["d_" + k] = sub(d, strides)
                rec["dl2_" + k] = np.float64(np.linalg.norm(d))
        else:
            g64 = rec  # fp32-vs-fp64 noise floor of the reference itself, per tensor (max over tensors reported)
            worst = 0.0
            for k, g in grads.items():
                ref = g64["gl2_" + k]
                if ref > 0:
                    worst = max(worst, abs(float(g.double().norm()) - ref) / ref)
            rec["g32_norm_dev_max"] = np.float64(worst)
    rec["meta"] = json.dumps(dict(case=name, cfg=cfg_name, B=B, T=T, t=[float(v) for v in t],
                                  mask=[bool(v) for v in mask], norm=norm, salt=salt, lr=lr_rate, wd=wd, clip=clip,
                                  full_limit=FULL_LIMIT, strides=strides, torch=torch.__version__, loss=loss_name,
                                  charbonnier_eps=charbonnier_eps,
                                  names=[k for k, _ in ref_model(cfg, norm, salt, torch.float32).named_parameters()]))
    np.savez_compressed(os.path.join(GOLD, f"train_{name}.npz"), **rec)
    sz = os.path.getsize(os.path.join(GOLD, f"train_{name}.npz"))
    print(f"[golden] train_{name}: loss={rec['loss64']:.6f} gnorm={rec['gnorm64']:.4f} "
          f"loss32-loss64={rec['loss32'] - rec['loss64']:.2e} ({sz / 1024:.0f} KiB)")


def big_case(name, cfg_name, B, T, t_list, mask_list, norm="rms", salt=0, clip=1.0, strides=(211, 97)):
    """Full-size (v3mod2: 766 M parameters, depth 28) step: loss, every parameter gradient (sub-sampled + full L2
    norms) and the clip norm of the reference model under fp64 autograd.  One precision and no optimiser state, so that
    the run fits the build container (model 6 GB + gradients 6 GB); the AdamW arithmetic is pinned by the small cases."""
    cfg = recipe.CONFIGS[cfg_name]
    hr, lr, noise = step_inputs(cfg, B, T, salt)
    t = np.asarray(t_list, dtype=np.float32)
    mask = np.asarray(mask_list, dtype=bool)
    dt = torch.float64
    m = ref_model(cfg, norm, salt, dt)
    hr_t, lr_t, nz = (torch.from_numpy(a).to(dt) for a in (hr, lr, noise))
    tt = torch.from_numpy(t).to(dt)
    lr_in = lr_t * (~torch.from_numpy(mask).view(B, 1, 1)).to(dt)
    tv = tt.view(-1, 1, 1)
    pred = m(tv * hr_t + (1 - tv) * nz, tt, lr_in)
    loss = torch.nn.functional.mse_loss(pred, hr_t)
    loss.backward()
    rec = {"loss64": np.float64(loss.item()), "pred_l2": np.float64(pred.detach().norm().item())}
    sq = 0.0
    names = []
    for k, p_ in m.named_parameters():
        names.append(k)
        g = p_.grad
        rec["g_" + k] = sub(g.numpy(), strides, stride1d=13)   # biases of 5120 / 7680 elements: ~500 samples, not 1
        n = float(g.norm().item())
        rec["gl2_" + k] = np.float64(n)
        sq += n * n
        p_.grad = None
    rec["gnorm64"] = np.float64(sq ** 0.5)     # what clip_grad_norm_ returns (train_ddp_v3m2.py:615)
    rec["meta"] = json.dumps(dict(case=name, cfg=cfg_name, B=B, T=T, t=[float(v) for v in t],
                                  mask=[bool(v) for v in mask], norm=norm, salt=salt, lr=5e-5, wd=0.1, clip=clip,
                                  full_limit=FULL_LIMIT, strides=strides, stride1d=13, torch=torch.__version__, names=names))
    np.savez_compressed(os.path.join(GOLD, f"train_{name}.npz"), **rec)
    sz = os.path.getsize(os.path.join(GOLD, f"train_{name}.npz"))
    print(f"[golden] train_{name}: loss={rec['loss64']:.6f} gnorm={rec['gnorm64']:.4f} ({sz / 1024:.0f} KiB)")


def dropout_case(name, cfg_name, B, T, t_list, seed, dropout=0.1, drop_path_rate=0.05, salt=0, strides=(7, 5)):
    """Train-mode Dropout / DropPath of the REFERENCE with injected masks.  torch's Philox stream cannot be reproduced
    on another device, so the masks come from the counter-based generator of csrc/jat_rng.h (numpy mirror
    oracle.jat_oracle_train.drop_mult) and are injected at the reference's own random calls, in call order:
    F.dropout(attn_weights) jat_audiosr_v3.py:175, torch.rand in drop_path :47 (attention branch, :300), F.dropout x2 in
    the MLP :269,271, torch.rand in drop_path (MLP branch, :306).  What is pinned is the SEMANTICS (where a mask applies,
    the 1/(1-p) scaling, how gradients flow through it)."""
    from oracle import jat_oracle_train as OT
    cfg = recipe.CONFIGS[cfg_name]
    depth, D, Hq = cfg["depth"], cfg["hidden_size"], cfg["num_q_heads"]
    mlp = int(D * cfg.get("mlp_ratio", 4.0))
    hr, lr, noise = step_inputs(cfg, B, T, salt)
    t = np.asarray(t_list, dtype=np.float32)
    N = -(-T // 4)
    dpr = [float(x) for x in torch.linspace(0, drop_path_rate, depth)]
    plan = OT.DropPlan(seed, [dropout] * depth, dpr)
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.jat_audiosr_v3 import JaT_AudioSR_V3
        m = JaT_AudioSR_V3(**cfg, dropout=dropout, drop_path_rate=drop_path_rate)
    sd = {k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg, "rms", salt).items()}
    m.load_state_dict(sd, strict=False)
    m = m.double().train()
    state = {"layer": 0, "drop_calls": 0, "rand_calls": 0}
    kinds_drop = [0, 2, 3]

    def fake_dropout(x, p=0.5, training=True, inplace=False):
        assert training and abs(p - dropout) < 1e-12
        layer, c = divmod(state["drop_calls"], 3)
        state["drop_calls"] += 1
        mult = plan.mult(layer, kinds_drop[c], tuple(x.shape))
        return x * torch.from_numpy(mult).to(x.dtype)

    def fake_rand(shape, dtype=None, device=None):
        # layers with drop_prob == 0 never call torch.rand (nn.Identity, :282); the others call it twice, in order
        live = [i for i in range(depth) if dpr[i] > 0]
        layer, c = live[state["rand_calls"] // 2], state["rand_calls"] % 2
        state["rand_calls"] += 1
        keep = plan.mult(layer, 1 if c == 0 else 4, (shape[0],)) > 0
        return torch.from_numpy(np.where(keep, 0.999, 0.0)).to(dtype).reshape(shape)   # floor(keep_prob + u) = keep

    real_drop, real_rand = torch.nn.functional.dropout, torch.rand
    torch.nn.functional.dropout, torch.rand = fake_dropout, fake_rand
    try:
        hr_t, lr_t, nz = (torch.from_numpy(a).double() for a in (hr, lr, noise))
        tt = torch.from_numpy(t).double()
        tv = tt.view(-1, 1, 1)
        z_t = tv * hr_t + (1 - tv) * nz
        pred = m(z_t, tt, lr_t)
        loss = torch.nn.functional.mse_loss(pred, hr_t)
        loss.backward()
    finally:
        torch.nn.functional.dropout, torch.rand = real_drop, real_rand
    assert state["drop_calls"] == 3 * depth and state["rand_calls"] == 2 * sum(1 for p in dpr if p > 0)
    rec = {"loss64": np.float64(loss.item()), "pred_l2": np.float64(pred.detach().norm().item())}
    for k, p in m.named_parameters():
        rec["g_" + k] = sub(p.grad.numpy(), strides)
        rec["gl2_" + k] = np.float64(p.grad.norm().item())
    rec["meta"] = json.dumps(dict(case=name, cfg=cfg_name, B=B, T=T, t=[float(v) for v in t], norm="rms", salt=salt,
                                  seed=seed, dropout=dropout, drop_path_rate=drop_path_rate, drop_path=dpr,
                                  full_limit=FULL_LIMIT, strides=strides, torch=torch.__version__,
                                  names=[k for k, _ in m.named_parameters()], mlp=mlp, N=N, Hq=Hq))
    np.savez_compressed(os.path.join(GOLD, f"train_{name}.npz"), **rec)
    print(f"[golden] train_{name}: loss={rec['loss64']:.6f} (dropout {dropout}, drop_path {dpr})")


def ref_loss_classes():
    """The four loss classes of train_ddp_v3mod2.py:53-321, taken from the file itself with `ast` (the module cannot be
    imported: tensorboard, process group)."""
    import torch.nn as nn
    import torch.nn.functional as F
    src = open(os.path.join(REF, "train_ddp_v3mod2.py"), encoding="utf-8").read()
    want = ("FrequencyDomainLatentLoss", "MultiScaleLatentLoss", "HybridConsistencyLoss", "CombinedLatentPerceptualLoss")
    body = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name in want]
    assert len(body) == 4
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(ast.Module(body=body, type_ignores=[]), "train_ddp_v3mod2.py", "exec"), ns)
    return ns["CombinedLatentPerceptualLoss"]


def loss_case(name, B, C, T, salt=0, lw=0.3, fw=0.5, mw=0.5, cw=0.1):
    """Value and d/d pred of  mse + lw * CombinedLatentPerceptualLoss(pred, target, lr)  (train_ddp_v3mod2.py:889-896,
    weights :362-366) on recipe tensors; fp32 inside, as the reference forces (`.float()`, :90-91)."""
    Loss = ref_loss_classes()
    fn = Loss(freq_weight=fw, ms_weight=mw, consistency_weight=cw, low_freq_phase_ratio=0.3)
    pred = torch.from_numpy(recipe.gaussian("loss_pred", (B, C, T), salt + 400)).requires_grad_(True)
    target = torch.from_numpy(recipe.gaussian("loss_target", (B, C, T), salt + 401))
    lr = torch.from_numpy(0.7 * recipe.gaussian("loss_target", (B, C, T), salt + 401)
                          + 0.5 * recipe.gaussian("loss_lr", (B, C, T), salt + 402)).float()
    mse = torch.nn.functional.mse_loss(pred, target)
    lat, terms = fn(pred, target, lr)
    total = mse + lw * lat
    total.backward()
    rec = {"total": np.float64(total.item()), "mse": np.float64(mse.item()), "freq": np.float64(terms["freq_loss"]),
           "ms": np.float64(terms["ms_loss"]), "consistency": np.float64(terms["consistency_loss"]),
           "latent": np.float64(terms["total_latent_loss"]), "dpred": pred.grad.numpy(),
           "meta": json.dumps(dict(case=name, B=B, C=C, T=T, salt=salt, lw=lw, fw=fw, mw=mw, cw=cw, torch=torch.__version__))}
    np.savez_compressed(os.path.join(GOLD, f"train_loss_{name}.npz"), **rec)
    print(f"[golden] train_loss_{name}: total={rec['total']:.6f} mse={rec['mse']:.6f} freq={rec['freq']:.5f} "
          f"ms={rec['ms']:.5f} cons={rec['consistency']:.5f}")


def mod2_step_case(name, cfg_name, B, T, t_list, salt=0, strides=(7, 5), lw=0.3, fw=0.5, mw=0.5, cw=0.1, stride1d=None):
    """One v3mod2 step end to end: JaT_AudioSR_V2 (LayerNorm) + MSE + latent perceptual loss with the clean LR latent
    (train_ddp_v3mod2.py:854-896), dropout = drop_path = 0, cond noise injected; gradients of every parameter (fp64
    model, the loss classes force fp32 internally exactly as in the trainer)."""
    Loss = ref_loss_classes()
    fn = Loss(freq_weight=fw, ms_weight=mw, consistency_weight=cw, low_freq_phase_ratio=0.3)
    cfg = recipe.CONFIGS[cfg_name]
    hr, lr, noise = step_inputs(cfg, B, T, salt)
    cnoise = 0.05 * recipe.gaussian("train_cnoise", hr.shape, salt + 303)
    t = np.asarray(t_list, dtype=np.float32)
    m = ref_model(cfg, "ln", salt, torch.float64)
    hr_t, lr_t, nz, cn = (torch.from_numpy(a).double() for a in (hr, lr, noise, cnoise))
    tt = torch.from_numpy(t).double()
    tv = tt.view(-1, 1, 1)
    pred = m(tv * hr_t + (1 - tv) * nz, tt, lr_t + cn)
    mse = torch.nn.functional.mse_loss(pred, hr_t)
    lat, terms = fn(pred, hr_t, lr_t)
    loss = mse + lw * lat
    loss.backward()
    rec = {"loss64": np.float64(loss.item()), "mse": np.float64(mse.item()), "latent": np.float64(lat.item()),
           "pred_l2": np.float64(pred.detach().norm().item())}
    for k, p in m.named_parameters():
        rec["g_" + k] = sub(p.grad.numpy(), strides, stride1d)
        rec["gl2_" + k] = np.float64(p.grad.norm().item())
    rec["meta"] = json.dumps(dict(case=name, cfg=cfg_name, B=B, T=T, t=[float(v) for v in t], norm="ln", salt=salt,
                                  lw=lw, fw=fw, mw=mw, cw=cw, full_limit=FULL_LIMIT, strides=strides, stride1d=stride1d,
                                  torch=torch.__version__, names=[k for k, _ in m.named_parameters()]))
    np.savez_compressed(os.path.join(GOLD, f"train_{name}.npz"), **rec)
    print(f"[golden] train_{name}: loss={rec['loss64']:.6f} (mse {rec['mse']:.6f} + {lw} * latent {rec['latent']:.5f})")


def u_shape_case():
    src = open(os.path.join(REF, "train_ddp_v3m2.py"), encoding="utf-8").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "u_shaped_timestep_sampling"]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "train_ddp_v3m2.py", "exec"), ns)
    u = recipe.uniform("u_shape", (64,), 5).astype(np.float32) * 0.5 + 0.5   # recipe.uniform is in [-1, 1)
    u[:4] = [0.0, 0.5, 0.25, 0.999999]
    real = torch.rand
    torch.rand = lambda *a, **k: torch.from_numpy(u).clone()
    try:
        t = ns["u_shaped_timestep_sampling"](64, "cpu").numpy()
    finally:
        torch.rand = real
    np.savez_compressed(os.path.join(GOLD, "train_misc.npz"), u=u, t_ushape=t)
    print(f"[golden] train_misc: t in [{t.min():.4f}, {t.max():.4f}]")


def main(which):
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 8)
    allc = not which
    if allc or "micro" in which:
        train_case("micro_T24", "micro", 2, 24, [0.1, 0.85], [False, True])
        train_case("micro_T22_pad", "micro", 3, 22, [0.02, 0.5, 0.97], [False, False, True], salt=1)
        train_case("micro_ln_T24", "micro", 2, 24, [0.3, 0.6], [False, False], norm="ln", salt=2)
    if allc or "tiny" in which:
        train_case("tiny_T128", "tiny", 2, 128, [0.2, 0.9], [False, True], strides=(61, 53))
        train_case("tiny_T1378", "tiny", 1, 1378, [0.4], [False], salt=1, strides=(61, 53))
    if allc or "drop" in which:
        dropout_case("micro_drop_T24", "micro", 3, 24, [0.1, 0.5, 0.85], seed=0x0123456789ABCDEF, dropout=0.1,
                     drop_path_rate=0.4)
        dropout_case("tiny_drop_T128", "tiny", 2, 128, [0.2, 0.9], seed=77, dropout=0.1, drop_path_rate=0.3,
                     strides=(61, 53))
    if allc or "loss" in which:
        loss_case("T24", 2, 32, 24)
        loss_case("T22", 3, 32, 22, salt=1)
        loss_case("T9", 1, 32, 9, salt=2)                 # odd T, empty / one-bin bands
        loss_case("T23", 2, 32, 23, salt=4)               # prime T: no factorisation, the direct-DFT kernel
        loss_case("T1378", 1, 64, 1378, salt=3)           # the trainer's crop: 690 bins, bands at 207 / 248
        mod2_step_case("micro_mod2_T24", "micro", 2, 24, [0.1, 0.85], salt=2)
        mod2_step_case("tiny_mod2_T128", "tiny", 2, 128, [0.2, 0.9], salt=1, strides=(61, 53))
        # well-conditioned variant (no log-magnitude term): the end-to-end v3mod2 gradient can be compared with the
        # reference's autograd directly (with fw > 0, d loss / d pred is sign(.)/(|P_k| + 1e-7): ill-conditioned)
        mod2_step_case("micro_mod2fw0_T24", "micro", 2, 24, [0.1, 0.85], salt=2, fw=0.0)
        mod2_step_case("tiny_mod2fw0_T128", "tiny", 2, 128, [0.2, 0.9], salt=1, strides=(61, 53), fw=0.0)
    if "v3mod2" in which:   # full-size model: ~15 GB of host memory, a few minutes; not part of the default set
        big_case("v3mod2_T128", "v3mod2", 2, 128, [0.2, 0.9], [False, True])
        big_case("v3mod2_T70_ragged", "v3mod2", 2, 70, [0.35, 0.8], [False, False], salt=1)
    if "v3mod2ln" in which:  # BASELINE configs[3]'s own combination at full size: JaT_AudioSR_V2 (LayerNorm, depth 28, 766 M parameters)
        # + MSE + latent perceptual loss (conditioned variant, fw = 0: comparable with reference autograd directly); ~20 GB, minutes
        mod2_step_case("v3mod2_mod2fw0_T128", "v3mod2", 2, 128, [0.2, 0.9], salt=1, strides=(211, 97), fw=0.0, stride1d=13)
    if allc or "charbonnier" in which:    # the V3M2-MOD1 trainer's reconstruction loss (train_ddp_v3m2mod1.py:72-101)
        charbonnier_case("T24", 2, 32, 24)
        charbonnier_case("T1378", 1, 8, 1378, salt=1)
        train_case("micro_charbonnier_T24", "micro", 2, 24, [0.1, 0.85], [False, True], salt=3, loss_name="charbonnier")
        train_case("tiny_charbonnier_T128", "tiny", 2, 128, [0.2, 0.9], [False, True], salt=2, strides=(61, 53),
                   loss_name="charbonnier")
    if allc or "misc" in which:
        u_shape_case()


if __name__ == "__main__":
    main(sys.argv[1:])
