"""CPU oracle for the JaTSR DiT sampling path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement (fp32 or fp64) of the reference algorithm on the hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product path
(`jatsr_amd`) never does and fails loudly when its HIP library is missing.

Pinned: `tests/test_oracle_golden.py` checks this file against fixtures under `tests/golden/` that were
produced by importing the reference's own classes (`oracle/gen_golden.py`, run in the build container where
/root/reference exists).  The reference itself holds no golden vectors for this path.

Each function cites the reference file:line it restates (paths relative to the reference root).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf as _erf


# --------------------------------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------------------------------
def gelu_erf(x):
    """nn.GELU() default (erf form) — src/models/jat_audiosr_v3.py:223,268."""
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype)


def silu(x):
    """nn.SiLU — src/models/jat_audiosr_v3.py:276,367."""
    return (x / (1.0 + np.exp(-x))).astype(x.dtype)


def linear(x, w, b=None):
    """nn.Linear: y = x W^T + b, W is [out,in] row-major."""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def rms_norm(x, weight, eps=1e-6):
    """nn.RMSNorm(D, eps=1e-6) — src/models/jat_audiosr_v3.py:261,264,384."""
    ms = np.mean(x * x, axis=-1, keepdims=True)
    y = x / np.sqrt(ms + eps)
    return y * weight if weight is not None else y


def layer_norm_noaffine(x, eps=1e-6):
    """nn.LayerNorm(D, elementwise_affine=False, eps=1e-6) — src/models/jat_audiosr_v2.py:242,245,361."""
    mu = np.mean(x, axis=-1, keepdims=True)
    var = np.mean((x - mu) ** 2, axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps)


def time_embedding(t, dim):
    """TimeEmbedding.forward — src/models/jat_audiosr_v3.py:194-207 (t in [0,1], unscaled)."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = np.exp(np.arange(half, dtype=np.float32) * np.float32(-k)).astype(t.dtype)
    e = t[:, None] * freqs[None, :]
    return np.concatenate([np.sin(e), np.cos(e)], axis=-1)


def rope_tables(head_dim, n, base=10000.0, dtype=np.float32):
    """RoPE.__init__ — src/models/jat_audiosr_v3.py:77-85 (tables are built in fp32 in the reference)."""
    inv_freq = (np.float32(1.0) / (np.float32(base) ** (np.arange(0, head_dim, 2, dtype=np.float32)
                                                          / np.float32(head_dim)))).astype(np.float32)
    t = np.arange(n, dtype=np.float32)
    freqs = np.outer(t, inv_freq).astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(dtype), np.sin(emb).astype(dtype)


def rotate_half(x):
    """RoPE.rotate_half — src/models/jat_audiosr_v3.py:104-108."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def apply_rope(x, cos, sin):
    """RoPE.forward on [B,N,H,hd] — src/models/jat_audiosr_v3.py:87-102."""
    return x * cos[None, :, None, :] + rotate_half(x) * sin[None, :, None, :]


def softmax_lastdim(s):
    m = np.max(s, axis=-1, keepdims=True)
    e = np.exp(s - m)
    return e / np.sum(e, axis=-1, keepdims=True)


# --------------------------------------------------------------------------------------------------
# modules
# --------------------------------------------------------------------------------------------------
class OracleModel:
    """JaT_AudioSR_V3 (norm='rms') / JaT_AudioSR_V2 (norm='ln') eval-mode forward in numpy.

    `sd` maps the reference state_dict key names to numpy arrays; RoPE buffers in `sd` are ignored and
    rebuilt (they are deterministic).  `dtype` selects fp32 (the reference's inference precision,
    infer_test_v3m2.py never enters autocast) or fp64 (ground truth for tolerance setting).
    """

    def __init__(self, cfg, sd, norm="rms", dtype=np.float32):
        self.cfg = dict(cfg)
        self.norm = norm
        self.dtype = dtype
        self.D = cfg["hidden_size"]
        self.depth = cfg["depth"]
        self.Hq = cfg["num_q_heads"]
        self.Hkv = cfg["num_kv_heads"]
        assert self.D % self.Hq == 0, "hidden_size must be divisible by num_q_heads"      # :119
        assert self.Hq % self.Hkv == 0, "num_q_heads must be divisible by num_kv_heads"   # :120
        self.hd = self.D // self.Hq
        self.P = cfg.get("patch_len", 4)
        self.Cin = cfg.get("input_channels", 1024)
        self.max_len = 2048                                                                # :361
        self.sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items() if ".rope." not in k}
        self.stages = None  # filled by forward(record=True)

    # -- sub-blocks -------------------------------------------------------------------------------
    def _norm(self, x, wkey):
        if self.norm == "rms":
            return rms_norm(x, self.sd[wkey])
        return layer_norm_noaffine(x)

    def patch_embed(self, x_in):
        """BottleneckPatchEmbed1D.forward — src/models/jat_audiosr_v3.py:229-248."""
        B, C, T = x_in.shape
        P = self.P
        x = x_in.reshape(B, C, T // P, P).transpose(0, 2, 1, 3).reshape(B, T // P, C * P)
        h = gelu_erf(linear(x, self.sd["patch_embed.proj.0.weight"], self.sd["patch_embed.proj.0.bias"]))
        return linear(h, self.sd["patch_embed.proj.2.weight"], self.sd["patch_embed.proj.2.bias"])

    def t_embed(self, t):
        """t_embedder — src/models/jat_audiosr_v3.py:364-369,455."""
        e = time_embedding(t.astype(self.dtype), self.D).astype(self.dtype)
        h = silu(linear(e, self.sd["t_embedder.1.weight"], self.sd["t_embedder.1.bias"]))
        return linear(h, self.sd["t_embedder.3.weight"], self.sd["t_embedder.3.bias"])

    def adaln(self, i, t_emb):
        """adaLN_modulation + chunk(6) — src/models/jat_audiosr_v3.py:275-278,293-294."""
        p = f"blocks.{i}.adaLN_modulation.1."
        return linear(silu(t_emb), self.sd[p + "weight"], self.sd[p + "bias"])

    def attention(self, i, x):
        """GroupedQueryAttention.forward (eval) — src/models/jat_audiosr_v3.py:144-184."""
        B, N, D = x.shape
        p = f"blocks.{i}.attn."
        Q = linear(x, self.sd[p + "q_proj.weight"]).reshape(B, N, self.Hq, self.hd)
        K = linear(x, self.sd[p + "k_proj.weight"]).reshape(B, N, self.Hkv, self.hd)
        V = linear(x, self.sd[p + "v_proj.weight"]).reshape(B, N, self.Hkv, self.hd)
        cos, sin = rope_tables(self.hd, N, dtype=self.dtype)
        Q = apply_rope(Q, cos, sin)
        K = apply_rope(K, cos, sin)
        g = self.Hq // self.Hkv
        K = np.repeat(K, g, axis=2)          # repeat_interleave: q-head h uses kv-head h // g  (:164-165)
        V = np.repeat(V, g, axis=2)
        Q, K, V = (a.transpose(0, 2, 1, 3) for a in (Q, K, V))
        S = (Q @ K.transpose(0, 1, 3, 2)) / math.sqrt(self.hd)
        O = softmax_lastdim(S) @ V
        O = O.transpose(0, 2, 1, 3).reshape(B, N, D)
        return linear(O, self.sd[p + "out_proj.weight"])

    def block(self, i, x, t_emb):
        """DiTBlock_GQA.forward (eval: drop_path/dropout are identity) — src/models/jat_audiosr_v3.py:284-308."""
        p = f"blocks.{i}."
        mod = self.adaln(i, t_emb)
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = np.split(mod, 6, axis=1)
        xn = self._norm(x, p + "norm1.weight") * (1 + sc_a[:, None, :]) + sh_a[:, None, :]
        x = x + g_a[:, None, :] * self.attention(i, xn)
        xn = self._norm(x, p + "norm2.weight") * (1 + sc_m[:, None, :]) + sh_m[:, None, :]
        h = gelu_erf(linear(xn, self.sd[p + "mlp.0.weight"], self.sd[p + "mlp.0.bias"]))
        x = x + g_m[:, None, :] * linear(h, self.sd[p + "mlp.3.weight"], self.sd[p + "mlp.3.bias"])
        return x

    def unpatchify(self, x, B, C, T):
        """unpatchify — src/models/jat_audiosr_v3.py:406-420."""
        N = x.shape[1]
        return x.reshape(B, N, C, self.P).transpose(0, 2, 1, 3).reshape(B, C, N * self.P)[:, :, :T]

    # -- whole forward ------------------------------------------------------------------------------
    def forward(self, x_t, t, x_cond, record=False):
        """JaT_AudioSR_V3.forward — src/models/jat_audiosr_v3.py:422-471."""
        x_t = np.asarray(x_t).astype(self.dtype)
        x_cond = np.asarray(x_cond).astype(self.dtype)
        t = np.asarray(t).astype(self.dtype)
        B, C, T_orig = x_t.shape
        P = self.P
        pad = (P - T_orig % P) % P
        if pad:
            x_t = np.pad(x_t, ((0, 0), (0, 0), (0, pad)))
            x_cond = np.pad(x_cond, ((0, 0), (0, 0), (0, pad)))
        T = x_t.shape[-1]
        x = self.patch_embed(np.concatenate([x_t, x_cond], axis=1))
        N = x.shape[1]
        if N > self.max_len:
            raise ValueError(f"Sequence length {N} exceeds max_len {self.max_len}")
        t_emb = self.t_embed(t)
        stages = {"patch_embed": x, "t_emb": t_emb} if record else None
        for i in range(self.depth):
            x = self.block(i, x, t_emb)
            if record:
                stages[f"block{i}"] = x
        xn = self._norm(x, "final_layer.0.weight")
        if record:
            stages["final_norm"] = xn
        y = linear(xn, self.sd["final_layer.1.weight"], self.sd["final_layer.1.bias"])
        out = self.unpatchify(y, B, self.Cin, T)[:, :, :T_orig]
        if record:
            self.stages = stages
        return out

    __call__ = forward


# --------------------------------------------------------------------------------------------------
# sampler + chunk driver
# --------------------------------------------------------------------------------------------------
def linspace_f32(a, b, n):
    """torch.linspace(a, b, n) in fp32 as PyTorch's CPU kernel computes it — infer_test_v3m2.py:136.
    step = fp32((b-a)/(n-1)); first half start + step*i, second half end - step*(n-1-i), each evaluated
    with ONE rounding (fused multiply-add), which is what the committed golden `linspace51` pins."""
    a32, b32 = np.float32(a), np.float32(b)
    step = float(np.float32((b32 - a32) / np.float32(n - 1)))
    out = np.empty(n, dtype=np.float32)
    half = n // 2
    for i in range(n):
        if i < half:
            out[i] = np.float32(float(a32) + step * i)          # exact in fp64, rounded once
        else:
            out[i] = np.float32(float(b32) - step * (n - 1 - i))
    return out


def flow_matching_sample(model, lr_latent, z0, num_steps=50, cfg_scale=1.0, max_steps=None):
    """flow_matching_sample — infer_test_v3m2.py:107-185, with the initial noise `z0` supplied by the
    caller instead of torch.randn (:133) so that results are reproducible across devices.
    max_steps: stop after that many Euler steps of the `num_steps` schedule (bench.py's time-boxed CPU baseline)."""
    dt_ = model.dtype
    lr = np.asarray(lr_latent).astype(dt_)
    z = np.asarray(z0).astype(dt_).copy()
    B = lr.shape[0]
    ts = linspace_f32(0.0, 1.0, num_steps + 1)
    use_cfg = cfg_scale != 1.0                                            # :139
    for i in range(num_steps if max_steps is None else min(num_steps, max_steps)):
        t_curr, t_next = ts[i], ts[i + 1]
        dt = np.float32(t_next - t_curr)
        tb = np.full((B,), t_curr, dtype=np.float32)
        if use_cfg:
            both = model.forward(np.concatenate([z, z], 0), np.concatenate([tb, tb], 0),
                                 np.concatenate([lr, np.zeros_like(lr)], 0))        # :154-158
            xc, xu = both[:B], both[B:]
            x_pred = xu + dt_(cfg_scale) * (xc - xu)                                # :164
        else:
            x_pred = model.forward(z, tb, lr)                                       # :167
        if t_curr < 0.999:                                                          # :173
            v = (x_pred - z) / dt_(np.float32(1) - t_curr + np.float32(1e-5))       # :175
            z = z + v * dt_(dt)                                                     # :176
        else:
            z = x_pred                                                              # :179
    return z


def crossfade_chunks(chunks, overlap_frames):
    """crossfade_chunks — infer_test_v3m2.py:188-233 (linear fade over `overlap_frames`)."""
    if len(chunks) == 0:
        return None
    if len(chunks) == 1:
        return chunks[0]
    result = chunks[0]
    for cur in chunks[1:]:
        if overlap_frames > 0 and result.shape[-1] >= overlap_frames:
            fo = np.linspace(1.0, 0.0, overlap_frames, dtype=np.float32).reshape(1, 1, -1)
            fi = np.linspace(0.0, 1.0, overlap_frames, dtype=np.float32).reshape(1, 1, -1)
            blended = result[..., -overlap_frames:] * fo + cur[..., :overlap_frames] * fi
            result = np.concatenate([result[..., :-overlap_frames], blended, cur[..., overlap_frames:]], -1)
        else:
            result = np.concatenate([result, cur], -1)
    return result


def chunk_plan(total_frames, chunk_frames=1378, overlap_frames=172):
    """Chunk boundaries — infer_test_v3m2.py:340-361,370-372 (16 s chunks, 2 s overlap at 44.1 kHz/512)."""
    stride = chunk_frames - overlap_frames
    num = (total_frames - overlap_frames + stride - 1) // stride
    return [(i * stride, min(i * stride + chunk_frames, total_frames)) for i in range(num)]


def cfg_euler_step(x_pred_2b, z, cfg_scale, t, dt):
    """One CFG combine + Euler update — infer_test_v3m2.py:161-179 (x_pred_2b = [cond; uncond])."""
    B = z.shape[0]
    xc, xu = x_pred_2b[:B], x_pred_2b[B:]
    x = xu + z.dtype.type(cfg_scale) * (xc - xu)
    if t < 0.999:
        return z + (x - z) / z.dtype.type(np.float32(1) - np.float32(t) + np.float32(1e-5)) * z.dtype.type(dt)
    return x


def forward_flops(cfg, B, T):
    """Algorithmic FLOPs of one forward (1 MAC = 2 FLOP), closed form of SURVEY.md §8d generalised to
    any config; equals torch.utils.flop_counter on the reference (checked in tests against the published
    127 627 689 984 per sample at T=512 for v3mod2 and 5 395 972 096 for tiny B=2,T=128)."""
    D = cfg["hidden_size"]; depth = cfg["depth"]; Hq = cfg["num_q_heads"]; Hkv = cfg["num_kv_heads"]
    hd = D // Hq; kvD = Hkv * hd; P = cfg.get("patch_len", 4)
    Cin = cfg.get("input_channels", 1024); Cc = cfg.get("cond_channels", 1024)
    bott = cfg.get("bottleneck_dim", 512); mlp = int(D * cfg.get("mlp_ratio", 4.0))
    N = -(-T // P)
    per_tok = 2 * (P * (Cin + Cc) * bott + bott * D)                     # patch embed
    per_tok += depth * 2 * (D * (D + 2 * kvD) + D * D + 2 * D * mlp)     # qkv, out, mlp
    per_tok += depth * 2 * (2 * N * D)                                   # QK^T and PV over N keys
    per_tok += 2 * D * P * Cin                                           # final linear
    per_sample = 2 * (2 * D * D) + depth * 2 * (D * 6 * D)               # t_embedder, adaLN
    return B * (N * per_tok + per_sample)
