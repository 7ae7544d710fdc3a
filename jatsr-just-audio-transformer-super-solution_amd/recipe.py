"""Deterministic synthetic weights and latents ("weights by recipe").

No trained checkpoint, dataset or stats file exists in the reference, and a freshly
constructed reference model outputs exact zeros because adaLN and the final linear are zero-initialised
(reference src/models/jat_audiosr_v3.py:395-404).  Every parity fixture and every bench run therefore
fills the model from this counter-based recipe: value = f(hash(parameter name), flat index).  It depends on
neither the PyTorch RNG nor the platform, so the reference class (in the build container) and this
package's model (on the GPU box) receive bit-identical fp32 weights and no weight file has to ship.

Pure numpy; not on the compute path.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _hash24(idx: np.ndarray, seed: np.uint64) -> np.ndarray:
    """24 well-mixed bits per element from (index, seed): one multiply-xorshift round on uint64 (wrap-around)."""
    with np.errstate(over="ignore"):
        h = idx * _GOLD
        h += seed
        h ^= h >> np.uint64(32)
        h *= _M1
        h ^= h >> np.uint64(29)
        h *= _M2
        h >>= np.uint64(40)
    return h


def name_seed(name: str, salt: int = 0) -> int:
    return (zlib.crc32(name.encode("utf-8")) | (salt << 32)) & 0xFFFFFFFFFFFFFFFF


_CHUNK = 1 << 20


def uniform(name: str, shape, salt: int = 0) -> np.ndarray:
    """Uniform (-1, 1) fp32 array addressed by (name, salt, flat index); exact IEEE fp32 arithmetic."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(name_seed(name, salt))
    out = np.empty(n, dtype=np.float32)
    for a in range(0, n, _CHUNK):          # chunked: stays in cache
        b = min(a + _CHUNK, n)
        u24 = _hash24(np.arange(a, b, dtype=np.uint64), seed).astype(np.float32)
        u24 *= np.float32(2.0 ** -23)
        u24 += np.float32(2.0 ** -24 - 1.0)
        out[a:b] = u24
    return out.reshape(shape)


def gaussian(name: str, shape, salt: int = 0) -> np.ndarray:
    """~N(0,1) fp32 array (Box-Muller over two hashed uniforms); used for latents and noise."""
    n = int(np.prod(shape))
    u1 = (uniform(name, (n,), salt * 2 + 1).astype(np.float64) + 1.0) * 0.5
    u2 = (uniform(name, (n,), salt * 2 + 2).astype(np.float64) + 1.0) * 0.5
    u1 = np.clip(u1, 2.0 ** -25, 1.0)
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return g.astype(np.float32).reshape(shape)


def model_param_shapes(cfg: dict, norm: str = "rms") -> "OrderedDict[str, tuple]":
    """state_dict parameter names/shapes of the reference model class for `cfg`.

    Order and names follow reference src/models/jat_audiosr_v3.py:350-386 (module registration order);
    the persistent RoPE buffers (`:78,84-85`) are produced by `rope_buffers`, not here.  `norm='ln'`
    gives the V2 class (LayerNorm without affine: no norm weights, jat_audiosr_v2.py:242,245,361).
    """
    D = cfg["hidden_size"]
    P = cfg.get("patch_len", 4)
    Cin = cfg.get("input_channels", 1024)
    Cc = cfg.get("cond_channels", 1024)
    bott = cfg.get("bottleneck_dim", 512)
    Hq, Hkv = cfg["num_q_heads"], cfg["num_kv_heads"]
    hd = D // Hq
    kvD = Hkv * hd
    mlp = int(D * cfg.get("mlp_ratio", 4.0))
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["patch_embed.proj.0.weight"] = (bott, P * (Cin + Cc))
    s["patch_embed.proj.0.bias"] = (bott,)
    s["patch_embed.proj.2.weight"] = (D, bott)
    s["patch_embed.proj.2.bias"] = (D,)
    s["t_embedder.1.weight"] = (D, D)
    s["t_embedder.1.bias"] = (D,)
    s["t_embedder.3.weight"] = (D, D)
    s["t_embedder.3.bias"] = (D,)
    for i in range(cfg["depth"]):
        p = f"blocks.{i}."
        if norm == "rms":
            s[p + "norm1.weight"] = (D,)
        s[p + "attn.q_proj.weight"] = (D, D)
        s[p + "attn.k_proj.weight"] = (kvD, D)
        s[p + "attn.v_proj.weight"] = (kvD, D)
        s[p + "attn.out_proj.weight"] = (D, D)
        if norm == "rms":
            s[p + "norm2.weight"] = (D,)
        s[p + "mlp.0.weight"] = (mlp, D)
        s[p + "mlp.0.bias"] = (mlp,)
        s[p + "mlp.3.weight"] = (D, mlp)
        s[p + "mlp.3.bias"] = (D,)
        s[p + "adaLN_modulation.1.weight"] = (6 * D, D)
        s[p + "adaLN_modulation.1.bias"] = (6 * D,)
    if norm == "rms":
        s["final_layer.0.weight"] = (D,)
    s["final_layer.1.weight"] = (P * Cin, D)
    s["final_layer.1.bias"] = (P * Cin,)
    return s


def make_param(name: str, shape, salt: int = 0) -> np.ndarray:
    """One parameter by recipe.  Linear weights/biases use the nn.Linear default bound 1/sqrt(fan_in);
    adaLN and final layers are NON-zero (fixture hazard, jat_audiosr_v3.py:395-404); norm weights != 1."""
    if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "final_layer.0.weight":
        return (1.0 + 0.25 * uniform(name, shape, salt)).astype(np.float32)
    if name.endswith(".weight"):
        fan_in = shape[1]
        bound = 1.0 / np.sqrt(fan_in)
        if "adaLN_modulation" in name:
            bound *= 1.5
        return (uniform(name, shape, salt) * np.float32(bound)).astype(np.float32)
    if name.endswith(".bias"):
        # fan_in of the owning Linear is not in the bias shape; a fixed modest bound keeps biases visible
        return (uniform(name, shape, salt) * np.float32(0.05)).astype(np.float32)
    raise KeyError(name)


def make_state_dict(cfg: dict, norm: str = "rms", salt: int = 0, threads: int = 8) -> "OrderedDict[str, np.ndarray]":
    shapes = model_param_shapes(cfg, norm)
    if threads <= 1:
        return OrderedDict((k, make_param(k, shp, salt)) for k, shp in shapes.items())
    from concurrent.futures import ThreadPoolExecutor   # numpy ufuncs release the GIL
    with ThreadPoolExecutor(threads) as ex:
        vals = list(ex.map(lambda kv: make_param(kv[0], kv[1], salt), shapes.items()))
    return OrderedDict(zip(shapes.keys(), vals))


def rope_buffers(head_dim: int, max_seq_len: int = 4096, base: float = 10000.0):
    """The reference's persistent RoPE buffers (jat_audiosr_v3.py:77-85), fp32 arithmetic throughout."""
    inv_freq = (np.float32(1.0) / (np.float32(base) ** (np.arange(0, head_dim, 2, dtype=np.float32)
                                                          / np.float32(head_dim)))).astype(np.float32)
    t = np.arange(max_seq_len, dtype=np.float32)
    freqs = np.outer(t, inv_freq).astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return inv_freq, np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)


def make_latents(B: int, C: int, T: int, salt: int = 0):
    """Synthetic normalised DAC latents: (x_t, x_cond) ~ N(0,1), shapes [B,C,T] fp32."""
    return gaussian("x_t", (B, C, T), salt), gaussian("x_cond", (B, C, T), salt)


def forward_flops(cfg: dict, B: int, T: int) -> int:
    """Algorithmic FLOPs of one DiT forward (1 MAC = 2 FLOP): the closed form of SURVEY.md §8d for any config,
    FLOPs = B * [N * (per-token GEMM + attention terms) + per-sample (t_embedder + adaLN) terms], N = ceil(T/4).
    Equal to torch.utils.flop_counter on the reference (tests/golden/misc.npz) — 127 627 689 984 per sample
    at T=512 for v3mod2.  bench.py prices every roofline fraction with this figure."""
    D = cfg["hidden_size"]; depth = cfg["depth"]; Hq = cfg["num_q_heads"]; Hkv = cfg["num_kv_heads"]
    hd = D // Hq; kvD = Hkv * hd; P = cfg.get("patch_len", 4)
    Cin = cfg.get("input_channels", 1024); Cc = cfg.get("cond_channels", 1024)
    bott = cfg.get("bottleneck_dim", 512); mlp = int(D * cfg.get("mlp_ratio", 4.0))
    N = -(-T // P)
    per_tok = 2 * (P * (Cin + Cc) * bott + bott * D)
    per_tok += depth * 2 * (D * (D + 2 * kvD) + D * D + 2 * D * mlp)
    per_tok += depth * 2 * (2 * N * D)
    per_tok += 2 * D * P * Cin
    per_sample = 2 * (2 * D * D) + depth * 2 * (D * 6 * D)
    return B * (N * per_tok + per_sample)


# Named configurations (BASELINE.json `configs`; SURVEY.md §8a sizes A/B; micro = <10 MB exact-check case)
CONFIGS = {
    "v3mod2": dict(input_channels=1024, cond_channels=1024, patch_len=4, hidden_size=1280, depth=28,
                   num_q_heads=20, num_kv_heads=4, bottleneck_dim=512, mlp_ratio=4.0),
    "tiny": dict(input_channels=1024, cond_channels=1024, patch_len=4, hidden_size=512, depth=12,
                 num_q_heads=8, num_kv_heads=4, bottleneck_dim=512, mlp_ratio=4.0),
    # v3mod2's layer dimensions (D=1280, 20Q/4KV, MLP 5120, 1024 channels) at depth 2: full-width parity cases that a
    # CPU oracle can afford
    "wide2": dict(input_channels=1024, cond_channels=1024, patch_len=4, hidden_size=1280, depth=2,
                  num_q_heads=20, num_kv_heads=4, bottleneck_dim=512, mlp_ratio=4.0),
    "micro": dict(input_channels=32, cond_channels=32, patch_len=4, hidden_size=256, depth=2,
                  num_q_heads=4, num_kv_heads=2, bottleneck_dim=128, mlp_ratio=4.0),
}
