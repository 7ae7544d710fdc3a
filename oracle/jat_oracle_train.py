"""CPU oracle of the TRAINING step — TEST INFRASTRUCTURE, never on the product path.

A numpy restatement, with hand-derived gradients (no autograd), of one optimisation step of the reference:

    pred = model(z_t, t, cond)          JaT_AudioSR_V3/V2.forward, train mode with dropout = drop_path = 0
                                         (src/models/jat_audiosr_v3.py:422-471; every sub-module cited below)
    loss = mean((pred - target)^2)      F.mse_loss                         train_ddp_v3m2.py:585
    g    = d loss / d params            loss.backward()                    :610
    g   *= min(1, clip/(||g|| + 1e-6))  clip_grad_norm_                    :614
    AdamW(lr, betas=(0.9,0.999), eps=1e-8, weight_decay)                   :423,618

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module.
PINNED: tests/test_train_cpu.py checks loss, every parameter gradient, the gradient norm and the AdamW parameter
deltas against tests/golden/train_*.npz, which oracle/gen_golden_train.py produced by running the reference's own model
class under torch autograd (fp64) in the build container.
"""
from __future__ import annotations

import math

import numpy as np

try:  # scipy is present in the image; fall back to math.erf for portability
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

from . import jat_oracle as O


def _dgelu(x):
    """d/dx of the erf-form GELU (nn.GELU default, jat_audiosr_v3.py:223,268)."""
    return 0.5 * (1.0 + _erf(x / math.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def _dsilu(x):
    s = 1.0 / (1.0 + np.exp(-x))
    return s * (1.0 + x * (1.0 - s))


# ---- dropout masks: numpy mirror of csrc/jat_rng.h (bit-exact uint32 arithmetic) ------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _hash32(x):
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & _M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & _M32
    x ^= x >> np.uint64(16)
    return x


def drop_mult(seed, site, p, shape):
    """Multipliers (0 or 1/(1-p)) of a mask site, elements numbered in C order of `shape` — jat_drop_spec/jat_drop_mult.
    site = layer*8 + kind; kind 0 attention probabilities [B,Hq,N,N], 1 DropPath(attn) [B], 2 MLP hidden [B,N,mlp],
    3 MLP output [B,N,D], 4 DropPath(MLP) [B]."""
    if p <= 0:
        return np.ones(shape)
    seed, site = int(seed), int(site)
    k0 = _hash32((seed & 0xFFFFFFFF) ^ ((site * 0x9e3779b9) & 0xFFFFFFFF))
    k1 = _hash32(((seed >> 32) + site * 0x85ebca6b + 1) & 0xFFFFFFFF)
    t = float(np.float32(p)) * 4294967296.0
    thresh = 4294967295 if t >= 4294967295.0 else int(t)
    idx = np.arange(int(np.prod(shape)), dtype=np.uint64)
    lo, hi = idx & _M32, idx >> np.uint64(32)
    rot = ((hi << np.uint64(13)) | (hi >> np.uint64(19))) & _M32
    r = _hash32(lo ^ k0 ^ rot) ^ k1
    inv_keep = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    return np.where(r < np.uint64(thresh), 0.0, inv_keep).reshape(shape)


class DropPlan:
    """Per-layer rates + step seed -> mask multipliers, as the HIP trainer draws them."""

    def __init__(self, seed, dropout, drop_path):
        self.seed, self.dropout, self.drop_path = seed, list(dropout), list(drop_path)

    def mult(self, layer, kind, shape):
        p = self.drop_path[layer] if kind in (1, 4) else self.dropout[layer]
        return drop_mult(self.seed, layer * 8 + kind, p, shape)


class TrainOracle:
    """fp64 forward with saved intermediates + explicit backward.  `sd`: reference state_dict names -> arrays."""

    def __init__(self, cfg, sd, norm="rms"):
        self.cfg, self.norm = dict(cfg), norm
        self.D, self.depth = cfg["hidden_size"], cfg["depth"]
        self.Hq, self.Hkv = cfg["num_q_heads"], cfg["num_kv_heads"]
        self.hd = self.D // self.Hq
        self.P, self.Cin = cfg.get("patch_len", 4), cfg.get("input_channels", 1024)
        self.sd = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items() if ".rope." not in k}

    # ---- norm (RMSNorm with weight: jat_audiosr_v3.py:261,264,384; LayerNorm no affine: jat_audiosr_v2.py:242,245,361) ----
    def _norm_fwd(self, x, wkey):
        if self.norm == "rms":
            rstd = 1.0 / np.sqrt(np.mean(x * x, -1, keepdims=True) + 1e-6)
            xh = x * rstd
            return xh * self.sd[wkey], (xh, rstd, self.sd[wkey])
        mu = np.mean(x, -1, keepdims=True)
        rstd = 1.0 / np.sqrt(np.mean((x - mu) ** 2, -1, keepdims=True) + 1e-6)
        xh = (x - mu) * rstd
        return xh, (xh, rstd, None)

    def _norm_bwd(self, dy, cache, wkey, grads):
        xh, rstd, w = cache
        g = dy * w if w is not None else dy
        if w is not None:
            grads[wkey] = (dy * xh).reshape(-1, xh.shape[-1]).sum(0)
        dx = g - xh * np.mean(g * xh, -1, keepdims=True)
        if self.norm != "rms":
            dx = dx - np.mean(g, -1, keepdims=True)
        return dx * rstd

    # ---- forward ----------------------------------------------------------------------------------------------------
    def forward(self, x_t, t, x_cond, plan=None):
        """plan: DropPlan (train-mode Dropout / DropPath with given masks) or None (both rates 0)."""
        sd, D, P = self.sd, self.D, self.P
        x_t, x_cond, t = (np.asarray(a, np.float64) for a in (x_t, x_cond, t))
        B, C, T0 = x_t.shape
        pad = (P - T0 % P) % P
        xin = np.pad(np.concatenate([x_t, x_cond], 1), ((0, 0), (0, 0), (0, pad)))          # :435-444
        N = xin.shape[-1] // P
        a0 = xin.reshape(B, xin.shape[1], N, P).transpose(0, 2, 1, 3).reshape(B, N, -1)      # :242-244
        c = {"B": B, "T0": T0, "N": N, "a0": a0}
        c["pe_pre"] = a0 @ sd["patch_embed.proj.0.weight"].T + sd["patch_embed.proj.0.bias"]  # :221-225
        c["pe_h"] = O.gelu_erf(c["pe_pre"])
        x = c["pe_h"] @ sd["patch_embed.proj.2.weight"].T + sd["patch_embed.proj.2.bias"]
        c["e"] = O.time_embedding(t, D)                                                          # :194-207
        c["u1"] = c["e"] @ sd["t_embedder.1.weight"].T + sd["t_embedder.1.bias"]                 # :364-369
        c["th"] = O.silu(c["u1"])
        c["temb"] = c["th"] @ sd["t_embedder.3.weight"].T + sd["t_embedder.3.bias"]
        c["st"] = O.silu(c["temb"])
        cos, sin = O.rope_tables(self.hd, N, dtype=np.float64)
        c["cos"], c["sin"] = cos, sin
        g = self.Hq // self.Hkv
        c["blocks"] = []
        for i in range(self.depth):
            p = f"blocks.{i}."
            k = {"x_in": x}
            mod = c["st"] @ sd[p + "adaLN_modulation.1.weight"].T + sd[p + "adaLN_modulation.1.bias"]   # :275-278
            k["mod"] = np.split(mod, 6, axis=1)
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = (m[:, None, :] for m in k["mod"])
            n1, k["n1c"] = self._norm_fwd(x, p + "norm1.weight")
            k["n1"] = n1
            k["xn1"] = n1 * (1 + sc_a) + sh_a                                                   # :297-298
            q = (k["xn1"] @ sd[p + "attn.q_proj.weight"].T).reshape(B, N, self.Hq, self.hd)      # :154-160
            kk = (k["xn1"] @ sd[p + "attn.k_proj.weight"].T).reshape(B, N, self.Hkv, self.hd)
            v = (k["xn1"] @ sd[p + "attn.v_proj.weight"].T).reshape(B, N, self.Hkv, self.hd)
            k["qr"] = O.apply_rope(q, cos, sin).transpose(0, 2, 1, 3)                            # [B,Hq,N,hd]
            k["kr"] = O.apply_rope(kk, cos, sin).transpose(0, 2, 1, 3)                           # [B,Hkv,N,hd]
            k["v"] = v.transpose(0, 2, 1, 3)
            Kx, Vx = np.repeat(k["kr"], g, axis=1), np.repeat(k["v"], g, axis=1)                 # :164-165
            k["p"] = O.softmax_lastdim(k["qr"] @ Kx.transpose(0, 1, 3, 2) / math.sqrt(self.hd))  # :167-174
            one = np.ones(())
            k["m_att"] = plan.mult(i, 0, k["p"].shape) if plan else one                         # self.dropout(attn_weights) :175
            k["m_pa"] = plan.mult(i, 1, (B,)).reshape(B, 1, 1) if plan else one                 # drop_path :300
            k["m_h"] = plan.mult(i, 2, (B, N, int(sd[p + "mlp.0.bias"].shape[0]))) if plan else one   # :269
            k["m_y"] = plan.mult(i, 3, (B, N, D)) if plan else one                              # :271
            k["m_pm"] = plan.mult(i, 4, (B,)).reshape(B, 1, 1) if plan else one                 # drop_path :306
            k["ao"] = ((k["p"] * k["m_att"]) @ Vx).transpose(0, 2, 1, 3).reshape(B, N, D)
            k["ya"] = k["ao"] @ sd[p + "attn.out_proj.weight"].T                                # :182
            x = x + k["m_pa"] * g_a * k["ya"]                                                   # :300
            k["x_mid"] = x
            n2, k["n2c"] = self._norm_fwd(x, p + "norm2.weight")
            k["n2"] = n2
            k["xn2"] = n2 * (1 + sc_m) + sh_m                                                   # :303-304
            k["hpre"] = k["xn2"] @ sd[p + "mlp.0.weight"].T + sd[p + "mlp.0.bias"]              # :266-270
            k["hpost"] = O.gelu_erf(k["hpre"]) * k["m_h"]
            k["ym"] = (k["hpost"] @ sd[p + "mlp.3.weight"].T + sd[p + "mlp.3.bias"]) * k["m_y"]
            x = x + k["m_pm"] * g_m * k["ym"]                                                   # :306
            c["blocks"].append(k)
        c["xf"] = x
        xn, c["nfc"] = self._norm_fwd(x, "final_layer.0.weight")                                # :383-386
        c["xnf"] = xn
        y = xn @ sd["final_layer.1.weight"].T + sd["final_layer.1.bias"]
        pred = y.reshape(B, N, self.Cin, P).transpose(0, 2, 1, 3).reshape(B, self.Cin, N * P)[:, :, :T0]   # :406-420
        self.cache = c
        return pred

    # ---- backward: returns {param name: gradient} for d loss / d param given d loss / d pred ------------------------------
    def backward(self, dpred):
        sd, D, P, c = self.sd, self.D, self.P, self.cache
        B, N, T0 = c["B"], c["N"], c["T0"]
        grads = {}
        dy = np.zeros((B, self.Cin, N * P))
        dy[:, :, :T0] = dpred
        dy = dy.reshape(B, self.Cin, N, P).transpose(0, 2, 1, 3).reshape(B, N, self.Cin * P)

        def lin_bwd(dyv, xv, wkey, bkey=None):
            grads[wkey] = dyv.reshape(-1, dyv.shape[-1]).T @ xv.reshape(-1, xv.shape[-1])
            if bkey:
                grads[bkey] = dyv.reshape(-1, dyv.shape[-1]).sum(0)
            return dyv @ sd[wkey]

        dxn = lin_bwd(dy, c["xnf"], "final_layer.1.weight", "final_layer.1.bias")
        dx = self._norm_bwd(dxn, c["nfc"], "final_layer.0.weight", grads)
        dst = np.zeros_like(c["st"])
        g = self.Hq // self.Hkv
        cos, sin = c["cos"][None, None], c["sin"][None, None]      # [1,1,N,hd]
        for i in reversed(range(self.depth)):
            p, k = f"blocks.{i}.", c["blocks"][i]
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = (m[:, None, :] for m in k["mod"])
            dmod = [None] * 6
            # x_out = x_mid + g_m * ym
            dmod[5] = (dx * k["ym"] * k["m_pm"]).sum(1)
            dh = lin_bwd(dx * g_m * k["m_pm"] * k["m_y"], k["hpost"], p + "mlp.3.weight", p + "mlp.3.bias") \
                * k["m_h"] * _dgelu(k["hpre"])
            dxn2 = lin_bwd(dh, k["xn2"], p + "mlp.0.weight", p + "mlp.0.bias")
            dmod[3], dmod[4] = dxn2.sum(1), (dxn2 * k["n2"]).sum(1)
            dx = dx + self._norm_bwd(dxn2 * (1 + sc_m), k["n2c"], p + "norm2.weight", grads)
            # x_mid = x_in + g_a * ya
            dmod[2] = (dx * k["ya"] * k["m_pa"]).sum(1)
            dao = lin_bwd(dx * g_a * k["m_pa"], k["ao"], p + "attn.out_proj.weight").reshape(B, N, self.Hq, self.hd).transpose(0, 2, 1, 3)
            Kx, Vx = np.repeat(k["kr"], g, axis=1), np.repeat(k["v"], g, axis=1)
            dV = (k["p"] * k["m_att"]).transpose(0, 1, 3, 2) @ dao
            dP = (dao @ Vx.transpose(0, 1, 3, 2)) * k["m_att"]
            dS = k["p"] * (dP - (dP * k["p"]).sum(-1, keepdims=True)) / math.sqrt(self.hd)
            dQ = dS @ Kx
            dK = dS.transpose(0, 1, 3, 2) @ k["qr"]
            dK = dK.reshape(B, self.Hkv, g, N, self.hd).sum(2)      # repeat_interleave^T
            dV = dV.reshape(B, self.Hkv, g, N, self.hd).sum(2)

            def rope_bwd(d):   # transpose of x*cos + rotate_half(x)*sin  (:87-108)
                h = self.hd // 2
                ds = d * sin
                return d * cos + np.concatenate([ds[..., h:], -ds[..., :h]], -1)

            dq = rope_bwd(dQ).transpose(0, 2, 1, 3).reshape(B, N, -1)
            dk = rope_bwd(dK).transpose(0, 2, 1, 3).reshape(B, N, -1)
            dv = dV.transpose(0, 2, 1, 3).reshape(B, N, -1)
            dxn1 = (lin_bwd(dq, k["xn1"], p + "attn.q_proj.weight") + lin_bwd(dk, k["xn1"], p + "attn.k_proj.weight")
                    + lin_bwd(dv, k["xn1"], p + "attn.v_proj.weight"))
            dmod[0], dmod[1] = dxn1.sum(1), (dxn1 * k["n1"]).sum(1)
            dx = dx + self._norm_bwd(dxn1 * (1 + sc_a), k["n1c"], p + "norm1.weight", grads)
            dst += lin_bwd(np.concatenate(dmod, 1), c["st"], p + "adaLN_modulation.1.weight", p + "adaLN_modulation.1.bias")
        dh1 = lin_bwd(dx, c["pe_h"], "patch_embed.proj.2.weight", "patch_embed.proj.2.bias") * _dgelu(c["pe_pre"])
        lin_bwd(dh1, c["a0"], "patch_embed.proj.0.weight", "patch_embed.proj.0.bias")
        dtemb = dst * _dsilu(c["temb"])
        du1 = lin_bwd(dtemb, c["th"], "t_embedder.3.weight", "t_embedder.3.bias") * _dsilu(c["u1"])
        lin_bwd(du1, c["e"], "t_embedder.1.weight", "t_embedder.1.bias")
        return grads

    # ---- the step ---------------------------------------------------------------------------------------------------
    def loss_and_grads(self, z_t, t, cond, target, plan=None, latent=None, cond_clean=None, charbonnier_eps=None):
        """latent: None (MSE, train_ddp_v3m2.py:585) or the keyword dict of latent_loss_oracle.latent_loss (the v3mod2
        loss, train_ddp_v3mod2.py:889-896, evaluated against the clean condition latent `cond_clean`);
        charbonnier_eps: the V3M2-MOD1 reconstruction loss instead of MSE (train_ddp_v3m2mod1.py:72-101,666-672)."""
        pred = self.forward(z_t, t, cond, plan)
        if charbonnier_eps is not None:
            loss, dpred = charbonnier_loss(pred, target, charbonnier_eps)
            return loss, self.backward(dpred), pred
        if latent:
            from . import latent_loss_oracle as LO
            terms, dpred = LO.latent_loss(pred, target, cond_clean, **latent)
            return float(terms["total"]), self.backward(dpred), pred
        diff = pred - np.asarray(target, np.float64)
        return float(np.mean(diff * diff)), self.backward(2.0 * diff / diff.size), pred

    @staticmethod
    def clip_and_adamw(params, grads, lr, weight_decay, max_norm=1.0, betas=(0.9, 0.999), eps=1e-8, step=1, state=None):
        """clip_grad_norm_ + one torch.optim.AdamW step.  Returns (grad_norm, {name: new param}, state)."""
        total = math.sqrt(sum(float((g * g).sum()) for g in grads.values()))
        coef = min(1.0, max_norm / (total + 1e-6)) if max_norm else 1.0
        state = state if state is not None else {}
        out = {}
        bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
        for k, p in params.items():
            g = grads[k] * coef
            m, v = state.get(k, (np.zeros_like(p), np.zeros_like(p)))
            m = betas[0] * m + (1 - betas[0]) * g
            v = betas[1] * v + (1 - betas[1]) * g * g
            state[k] = (m, v)
            out[k] = p * (1 - lr * weight_decay) - (lr / bc1) * m / (np.sqrt(v) / math.sqrt(bc2) + eps)
        return total, out, state


def charbonnier_loss(pred, target, eps=1e-6):
    """train_ddp_v3m2mod1.py:72-101: mean(sqrt((pred - target)^2 + eps)) — eps is added to the SQUARED difference — and its
    gradient d / sqrt(d^2 + eps) / n.  Returns (loss, dpred) in fp64."""
    d = np.asarray(pred, np.float64) - np.asarray(target, np.float64)
    r = np.sqrt(d * d + eps)
    return float(r.mean()), d / r / d.size


def u_shaped_timestep_sampling(u, alpha=0.5):
    """train_ddp_v3m2.py:164-172 on given uniform draws."""
    u = np.asarray(u)
    return np.where(u < 0.5, (2 * u) ** alpha / 2, 1 - ((2 * (1 - u)) ** alpha) / 2).astype(u.dtype)
