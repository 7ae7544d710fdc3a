"""CPU oracle of the v3mod2 trainer's loss — TEST INFRASTRUCTURE, never on the product path.

numpy restatement (values AND hand-derived gradient w.r.t. the prediction) of
    CombinedLatentPerceptualLoss                         train_ddp_v3mod2.py:274-321
      FrequencyDomainLatentLoss   log-magnitude L1 + 0.1 * low-band complex L1    :53-123
      MultiScaleLatentLoss        L1 at scales 1, 2, 4 (AvgPool1d), averaged      :126-171
      HybridConsistencyLoss       strict-band complex L1 + decayed magnitude L1   :174-271
    loss = mse + latent_weight * (fw * freq + mw * ms + cw * cons)                 :889-896
PINNED: tests/test_train_cpu.py compares values and gradients with tests/golden/train_loss_*.npz, produced by
oracle/gen_golden_train.py from the reference's own classes (AST-extracted) under torch autograd.
"""
from __future__ import annotations

import numpy as np


def _dft_mats(T):
    F = T // 2 + 1
    th = 2.0 * np.pi * np.outer(np.arange(F), np.arange(T)) / T
    return np.cos(th), np.sin(th)


def latent_loss(pred, target, lr, latent_weight=0.3, freq_weight=0.5, ms_weight=0.5, consistency_weight=0.1,
                low_freq_phase_ratio=0.3, strict_cutoff=0.30, soft_cutoff=0.36):
    """-> (terms dict, d total / d pred).  pred, target, lr: [B, C, T]."""
    p, h, r = (np.asarray(a, np.float64) for a in (pred, target, lr))
    B, C, T = p.shape
    rows = B * C
    P, H, R = (np.fft.rfft(a, axis=-1) for a in (p, h, r))
    F = P.shape[-1]
    low, strict, soft = int(F * low_freq_phase_ratio), int(F * strict_cutoff), int(F * soft_cutoff)   # :111,232-233
    eps = 1e-7
    g = np.zeros_like(P)          # ga + i gb = dL/dRe(P) + i dL/dIm(P), already weighted inside the latent sum
    pm, hm, rm = np.abs(P), np.abs(H), np.abs(R)
    safe = np.where(pm > 0, pm, 1.0)
    unit = np.where(pm > 0, P / safe, 0.0)
    # 1. log-magnitude L1 (:97-107)
    d = np.log(pm + eps) - np.log(hm + eps)
    log_mag = np.abs(d).mean()
    g += freq_weight * np.sign(d) / (pm + eps) / (rows * F) * unit
    # 2. low-band complex L1 (:109-117), weight 0.1 (:121)
    low_loss = 0.0
    if low > 0:
        dl = P[..., :low] - H[..., :low]
        m = np.abs(dl)
        low_loss = m.mean()
        g[..., :low] += freq_weight * 0.1 * np.where(m > 0, dl / np.where(m > 0, m, 1.0), 0.0) / (rows * low)
    freq = log_mag + 0.1 * low_loss
    # 3. consistency (:229-262)
    strict_loss = trans_loss = 0.0
    if strict > 0:
        ds = P[..., :strict] - R[..., :strict]
        m = np.abs(ds)
        strict_loss = m.mean()
        g[..., :strict] += consistency_weight * np.where(m > 0, ds / np.where(m > 0, m, 1.0), 0.0) / (rows * strict)
    bw = soft - strict
    if bw > 0:
        w = np.linspace(1.0, 0.0, bw)                       # torch.linspace(1.0, 0.0, steps=band_width) :251
        dt = pm[..., strict:soft] - rm[..., strict:soft]
        trans_loss = (np.abs(dt) * w).mean()
        g[..., strict:soft] += consistency_weight * w * np.sign(dt) / (rows * bw) * unit[..., strict:soft]
    cons = strict_loss + trans_loss
    # adjoint of rfft restricted to the F bins: dp_n = sum_k ga cos(theta) - gb sin(theta)
    Cm, Sm = _dft_mats(T)
    dspec = g.real @ Cm - g.imag @ Sm
    # 4. multi-scale L1 (:158-171), in the time domain
    e = p - h
    ms_sum = np.abs(e).mean()
    dms = np.sign(e) / e.size
    for s in (2, 4):
        Ts = T // s
        if Ts == 0:
            continue
        q = e[..., :Ts * s].reshape(B, C, Ts, s).mean(-1)
        ms_sum += np.abs(q).mean()
        dms[..., :Ts * s] += np.repeat(np.sign(q), s, axis=-1) / s / q.size
    ms = ms_sum / 3.0
    mse = (e * e).mean()
    latent = freq_weight * freq + ms_weight * ms + consistency_weight * cons
    total = mse + latent_weight * latent
    dpred = 2.0 * e / e.size + latent_weight * (dspec + ms_weight * dms / 3.0)
    return dict(total=total, mse=mse, freq=freq, ms=ms, consistency=cons, latent=latent), dpred
