"""Host-side twins of the reference model classes.

Same class names, constructor keywords, `forward` signatures, `state_dict` keys and error behaviour as
reference src/models/jat_audiosr_v3.py (and `_v2.py` for the LayerNorm variant), so that reference
checkpoints load unchanged (`infer_test_v3m2.py:61-74`) and reference-style call sites keep working.  The
modules below only HOLD the fp32 parameters (PyTorch tensors are containers); every forward dispatches
to the gfx950 kernels behind the C ABI in include/jat_hip.h.  Inference (eval) semantics only: dropout and
DropPath are identity (jat_audiosr_v3.py:45,139,269-271), and nothing here is differentiable.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref

import torch
import torch.nn as nn

from . import _lib as L


def _rope_buffers(head_dim, max_seq_len=4096, base=10000):
    """Persistent buffers of the reference RoPE module (jat_audiosr_v3.py:77-85); kept only so that
    state_dict() has the reference's keys — the kernels use their own fp32 table."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_seq_len).float()
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat([freqs, freqs], dim=-1)
    return inv_freq, emb.cos(), emb.sin()


class RoPE(nn.Module):
    """Holder of the reference's RoPE buffers (jat_audiosr_v3.py:67-108); rotation happens in the QKV GEMM epilogue."""

    def __init__(self, dim, max_seq_len=4096, base=10000):
        super().__init__()
        self.dim, self.max_seq_len, self.base = dim, max_seq_len, base
        inv_freq, cos, sin = _rope_buffers(dim, max_seq_len, base)
        self.register_buffer("inv_freq", inv_freq)
        self.register_buffer("cos_cached", cos)
        self.register_buffer("sin_cached", sin)


class _NormHolder(nn.Module):
    """nn.RMSNorm(D, eps=1e-6) parameter holder (weight only) — jat_audiosr_v3.py:261,264,384."""

    def __init__(self, dim, affine=True):
        super().__init__()
        self.eps = 1e-6
        if affine:
            self.weight = nn.Parameter(torch.ones(dim))


class _Handle:
    """Owns one `jat_model*` plus its packed-weights freshness and a cached workspace tensor."""

    def __init__(self, cfg: dict, norm_mode: int):
        self.cfg = cfg
        self.norm_mode = norm_mode
        self.ptr = C.c_void_p()
        self.version = None
        self.epoch = 0          # bumped when a trainer updates the weights in place (invalidates cached samplers)
        self.trainer = None     # weakref to the attached jatsr_amd.Trainer (its transposed operand copies follow a re-pack)
        self.device = None
        self._ws = None
        c = L.JatConfig(cfg["input_channels"], cfg["cond_channels"], cfg["patch_len"], cfg["hidden_size"],
                        cfg["depth"], cfg["num_q_heads"], cfg["num_kv_heads"], cfg["bottleneck_dim"],
                        cfg["mlp_hidden"], norm_mode)
        L.check(L.lib().jat_model_create(C.byref(c), C.byref(self.ptr)))

    def __del__(self):
        try:
            if self.ptr:
                L.lib().jat_model_destroy(self.ptr)
        except Exception:
            pass

    def load(self, named: dict):
        """named: reference state_dict key -> fp32 CUDA tensor."""
        L.require_gpu()
        keep = []
        refs = (L.JatTensorRef * len(named))()
        for i, (k, v) in enumerate(named.items()):
            if v.dtype != torch.float32 or not v.is_cuda:
                raise L.JatError(f"parameter {k} must be an fp32 CUDA tensor (got {v.dtype} on {v.device})")
            v = v.detach().contiguous()
            keep.append(v)
            refs[i] = L.JatTensorRef(k.encode(), v.data_ptr(), v.numel())
        L.check(L.lib().jat_model_load_weights(self.ptr, refs, len(named), L.stream_ptr()))

    def set_switch(self, name: str, value: int):
        """Per-handle behaviour switch (jat_model_set_switch): the JAT_* environment variables only set the defaults, once, when the
        handle is created.  Applies to forwards enqueued and samplers created afterwards."""
        L.check(L.lib().jat_model_set_switch(self.ptr, name.encode(), int(value)))

    def workspace(self, B, T, device):
        need = C.c_size_t()
        L.check(L.lib().jat_model_workspace_bytes(self.ptr, B, T, C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws


def _params_version(module: nn.Module):
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


class _PackedMixin:
    """Lazily (re)packs the module's fp32 parameters into the C-side bf16 layout when they change."""

    def _handle_cfg(self):
        raise NotImplementedError

    def _named_for_pack(self):
        raise NotImplementedError

    def _get_handle(self) -> _Handle:
        L.require_gpu()
        h = self.__dict__.get("_jat_handle")
        if h is None:
            cfg, mode = self._handle_cfg()
            h = _Handle(cfg, mode)
            self.__dict__["_jat_handle"] = h
        ver = _params_version(self)
        if h.version != ver:
            named = self._named_for_pack()
            dev = next(iter(named.values())).device
            if dev.type != "cuda":
                raise L.JatError("move the model to the GPU first (.to('cuda')); there is no CPU fallback")
            h.load(named)
            h.version = ver
            h.device = dev
            tr = h.trainer() if h.trainer is not None else None
            if tr is not None:        # e.g. model.load_state_dict() while a Trainer is attached: its W^T copies are stale too
                tr._weights_replaced()
        return h


def _check_f32_cuda(*tensors):
    L.require_gpu()
    out = []
    for t in tensors:
        if not t.is_cuda:
            raise L.JatError("inputs must be CUDA tensors; there is no CPU fallback")
        out.append(t.detach().to(torch.float32).contiguous())
    return out


class GroupedQueryAttention(nn.Module, _PackedMixin):
    """GQA with RoPE — reference jat_audiosr_v3.py:111-184.  forward(x[B,N,D]) -> [B,N,D]."""

    def __init__(self, hidden_size, num_q_heads, num_kv_heads, dropout=0.0):
        super().__init__()
        assert hidden_size % num_q_heads == 0, "hidden_size must be divisible by num_q_heads"
        assert num_q_heads % num_kv_heads == 0, "num_q_heads must be divisible by num_kv_heads"
        self.hidden_size = hidden_size
        self.num_q_heads = num_q_heads
        self.num_kv_heads = num_kv_heads
        self.num_groups = num_q_heads // num_kv_heads
        self.head_dim = hidden_size // num_q_heads
        kv = self.num_kv_heads * self.head_dim
        self.q_proj = nn.Linear(hidden_size, hidden_size, bias=False)
        self.k_proj = nn.Linear(hidden_size, kv, bias=False)
        self.v_proj = nn.Linear(hidden_size, kv, bias=False)
        self.out_proj = nn.Linear(hidden_size, hidden_size, bias=False)
        self.dropout = nn.Dropout(dropout)
        self.rope = RoPE(self.head_dim)
        self._owner = None  # (weakref to parent block/model handle provider, layer index)

    # standalone use: a 1-layer handle whose other weights are dummies
    def _handle_cfg(self):
        return _standalone_cfg(self.hidden_size, self.num_q_heads, self.num_kv_heads, self.hidden_size * 4), L.NORM_RMS_W

    def _named_for_pack(self):
        dev = self.q_proj.weight.device
        named = _dummy_named(self._handle_cfg()[0], dev)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            named[f"blocks.0.attn.{n}.weight"] = getattr(self, n).weight
        return named

    def forward(self, x):
        (x,) = _check_f32_cuda(x)
        B, N, D = x.shape
        owner = self._owner() if self._owner is not None else None
        if owner is not None:
            h, layer = owner._get_handle(), self._layer
        else:
            h, layer = self._get_handle(), 0
        y = torch.empty_like(x)
        ws = h.workspace(B, 4 * N, x.device)
        L.check(L.lib().jat_attn_forward(h.ptr, layer, L.ptr(x), L.ptr(y), B, N, L.ptr(ws), ws.numel(), L.stream_ptr()))
        return y


class DiTBlock_GQA(nn.Module, _PackedMixin):
    """DiT block with GQA and adaLN-Zero — reference jat_audiosr_v3.py:251-308.
    forward(x[B,N,D], t_emb[B,D]) -> [B,N,D].  `norm='ln'` gives the V2 block (LayerNorm, no affine)."""

    def __init__(self, hidden_size, num_q_heads, num_kv_heads, mlp_ratio=4.0, dropout=0.1, drop_path=0.0, norm="rms"):
        super().__init__()
        self._norm_kind = norm
        self.norm1 = _NormHolder(hidden_size, affine=(norm == "rms"))
        self.attn = GroupedQueryAttention(hidden_size, num_q_heads, num_kv_heads, dropout=dropout)
        self.norm2 = _NormHolder(hidden_size, affine=(norm == "rms"))
        mlp_hidden_dim = int(hidden_size * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(hidden_size, mlp_hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(mlp_hidden_dim, hidden_size), nn.Dropout(dropout))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))
        self.drop_path = nn.Identity()  # eval-mode identity (jat_audiosr_v3.py:45)
        self.dropout_rate, self.drop_path_rate = float(dropout), float(drop_path)   # training-only (jatsr_amd.train)
        self.hidden_size, self.num_q_heads, self.num_kv_heads = hidden_size, num_q_heads, num_kv_heads
        self.mlp_hidden = mlp_hidden_dim
        self._owner = None
        self._layer = 0
        self.attn._owner = weakref.ref(self)
        self.attn._layer = 0

    def _handle_cfg(self):
        mode = L.NORM_RMS_W if self._norm_kind == "rms" else L.NORM_LN_NOAFFINE
        return _standalone_cfg(self.hidden_size, self.num_q_heads, self.num_kv_heads, self.mlp_hidden), mode

    def _named_for_pack(self):
        dev = self.mlp[0].weight.device
        named = _dummy_named(self._handle_cfg()[0], dev, norm=self._norm_kind)
        for k, v in self.state_dict(keep_vars=True).items():
            if ".rope." not in k:
                named["blocks.0." + k] = v
        return named

    def _get_handle(self):
        owner = self._owner() if self._owner is not None else None
        if owner is not None:
            return owner._get_handle()
        return _PackedMixin._get_handle(self)

    def forward(self, x, t_emb):
        x, t_emb = _check_f32_cuda(x, t_emb)
        B, N, D = x.shape
        h = self._get_handle()
        layer = self._layer if (self._owner is not None and self._owner() is not None) else 0
        y = torch.empty_like(x)
        ws = h.workspace(B, 4 * N, x.device)
        L.check(L.lib().jat_block_forward(h.ptr, layer, L.ptr(x), L.ptr(t_emb), L.ptr(y), B, N, L.ptr(ws), ws.numel(),
                                          L.stream_ptr()))
        return y


class TimeEmbedding(nn.Module):
    """Parameter-free sinusoidal embedding (jat_audiosr_v3.py:187-207); computed inside jat_time_embed."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim


class BottleneckPatchEmbed1D(nn.Module):
    """Parameter holder for the bottleneck patch embedding (jat_audiosr_v3.py:210-248)."""

    def __init__(self, patch_len, in_chans, embed_dim, bottleneck_dim):
        super().__init__()
        self.patch_len = patch_len
        self.flatten_dim = patch_len * in_chans
        self.proj = nn.Sequential(nn.Linear(self.flatten_dim, bottleneck_dim), nn.GELU(),
                                  nn.Linear(bottleneck_dim, embed_dim))


def _standalone_cfg(D, Hq, Hkv, mlp_hidden):
    return dict(input_channels=32, cond_channels=32, patch_len=4, hidden_size=D, depth=1, num_q_heads=Hq,
                num_kv_heads=Hkv, bottleneck_dim=128, mlp_hidden=mlp_hidden)


def _dummy_named(cfg, device, norm="rms"):
    """Zero tensors for every parameter a 1-layer handle expects (standalone block / attention use)."""
    D, bott, mlp = cfg["hidden_size"], cfg["bottleneck_dim"], cfg["mlp_hidden"]
    kv = cfg["num_kv_heads"] * 64
    Kp = 4 * (cfg["input_channels"] + cfg["cond_channels"])
    shapes = {
        "patch_embed.proj.0.weight": (bott, Kp), "patch_embed.proj.0.bias": (bott,),
        "patch_embed.proj.2.weight": (D, bott), "patch_embed.proj.2.bias": (D,),
        "t_embedder.1.weight": (D, D), "t_embedder.1.bias": (D,), "t_embedder.3.weight": (D, D),
        "t_embedder.3.bias": (D,),
        "blocks.0.attn.q_proj.weight": (D, D), "blocks.0.attn.k_proj.weight": (kv, D),
        "blocks.0.attn.v_proj.weight": (kv, D), "blocks.0.attn.out_proj.weight": (D, D),
        "blocks.0.mlp.0.weight": (mlp, D), "blocks.0.mlp.0.bias": (mlp,), "blocks.0.mlp.3.weight": (D, mlp),
        "blocks.0.mlp.3.bias": (D,), "blocks.0.adaLN_modulation.1.weight": (6 * D, D),
        "blocks.0.adaLN_modulation.1.bias": (6 * D,),
        "final_layer.1.weight": (4 * cfg["input_channels"], D), "final_layer.1.bias": (4 * cfg["input_channels"],),
    }
    if norm == "rms":
        shapes.update({"blocks.0.norm1.weight": (D,), "blocks.0.norm2.weight": (D,), "final_layer.0.weight": (D,)})
    return {k: torch.zeros(s, dtype=torch.float32, device=device) for k, s in shapes.items()}


class JaT_AudioSR_V3(nn.Module, _PackedMixin):
    """JaT-AudioSR V3 (RMSNorm) — reference jat_audiosr_v3.py:311-471.
    forward(x_t[B,C,T], t[B], x_cond[B,C,T]) -> x_pred[B,C,T]."""

    _NORM = "rms"

    def __init__(self, input_channels=1024, cond_channels=1024, patch_len=4, hidden_size=1024, depth=16,
                 num_q_heads=16, num_kv_heads=4, bottleneck_dim=512, mlp_ratio=4.0, dropout=0.1,
                 drop_path_rate=0.0):
        super().__init__()
        self.input_channels = input_channels
        self.cond_channels = cond_channels
        self.patch_len = patch_len
        self.hidden_size = hidden_size
        self.depth = depth
        self.num_q_heads, self.num_kv_heads = num_q_heads, num_kv_heads
        self.bottleneck_dim = bottleneck_dim
        self.mlp_hidden = int(hidden_size * mlp_ratio)
        self.patch_embed = BottleneckPatchEmbed1D(patch_len, input_channels + cond_channels, hidden_size, bottleneck_dim)
        self.max_len = 2048
        self.t_embedder = nn.Sequential(TimeEmbedding(hidden_size), nn.Linear(hidden_size, hidden_size), nn.SiLU(),
                                        nn.Linear(hidden_size, hidden_size))
        self.blocks = nn.ModuleList([
            DiTBlock_GQA(hidden_size, num_q_heads, num_kv_heads, mlp_ratio, dropout=dropout, norm=self._NORM,
                         drop_path=(drop_path_rate * i / (depth - 1) if depth > 1 else 0.0))   # linspace(0, rate, depth), :372-377
            for i in range(depth)])
        patch_out_dim = patch_len * input_channels
        if self._NORM == "rms":
            self.final_layer = nn.Sequential(_NormHolder(hidden_size), nn.Linear(hidden_size, patch_out_dim))
        else:
            self.final_layer = nn.Sequential(_NormHolder(hidden_size, affine=False), nn.Linear(hidden_size, patch_out_dim))
        for i, blk in enumerate(self.blocks):
            blk._owner = weakref.ref(self)
            blk._layer = i
            blk.attn._owner = weakref.ref(self)
            blk.attn._layer = i
        self.initialize_weights()

    def initialize_weights(self):
        """Zero adaLN modulation and the final linear (jat_audiosr_v3.py:395-404): a fresh model outputs 0."""
        for block in self.blocks:
            nn.init.constant_(block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer[-1].weight, 0)
        nn.init.constant_(self.final_layer[-1].bias, 0)

    # -- C-side handle ---------------------------------------------------------------------------------
    def _handle_cfg(self):
        cfg = dict(input_channels=self.input_channels, cond_channels=self.cond_channels, patch_len=self.patch_len,
                   hidden_size=self.hidden_size, depth=self.depth, num_q_heads=self.num_q_heads,
                   num_kv_heads=self.num_kv_heads, bottleneck_dim=self.bottleneck_dim, mlp_hidden=self.mlp_hidden)
        return cfg, (L.NORM_RMS_W if self._NORM == "rms" else L.NORM_LN_NOAFFINE)

    def _named_for_pack(self):
        return {k: v for k, v in self.state_dict(keep_vars=True).items() if ".rope." not in k}

    def config(self):
        return dict(input_channels=self.input_channels, cond_channels=self.cond_channels, patch_len=self.patch_len,
                    hidden_size=self.hidden_size, depth=self.depth, num_q_heads=self.num_q_heads,
                    num_kv_heads=self.num_kv_heads, bottleneck_dim=self.bottleneck_dim,
                    mlp_ratio=self.mlp_hidden / self.hidden_size)

    # -- reference API -----------------------------------------------------------------------------------
    def forward(self, x_t, t, x_cond):
        x_t, t, x_cond = _check_f32_cuda(x_t, t, x_cond)
        B, C_, T = x_t.shape
        if C_ != self.input_channels or x_cond.shape != (B, self.cond_channels, T) or t.shape != (B,):
            raise ValueError(f"shape mismatch: x_t {tuple(x_t.shape)}, t {tuple(t.shape)}, x_cond {tuple(x_cond.shape)}")
        N = math.ceil(T / self.patch_len)
        if N > self.max_len:
            raise ValueError(f"Sequence length {N} exceeds max_len {self.max_len}")  # jat_audiosr_v3.py:451-452
        h = self._get_handle()
        out = torch.empty_like(x_t)
        ws = h.workspace(B, T, x_t.device)
        L.check(L.lib().jat_forward(h.ptr, L.ptr(x_t), L.ptr(t), L.ptr(x_cond), L.ptr(out), B, T, L.ptr(ws), ws.numel(),
                                    L.stream_ptr()))
        return out

    def time_embed(self, t):
        """t_embedder(t) (jat_audiosr_v3.py:455) -> [B, D]."""
        (t,) = _check_f32_cuda(t)
        h = self._get_handle()
        B = t.shape[0]
        out = torch.empty(B, self.hidden_size, dtype=torch.float32, device=t.device)
        ws = h.workspace(B, 4, t.device)
        L.check(L.lib().jat_time_embed(h.ptr, L.ptr(t), L.ptr(out), B, L.ptr(ws), ws.numel(), L.stream_ptr()))
        return out


class JaT_AudioSR_V2(JaT_AudioSR_V3):
    """The LayerNorm(no affine) twin — reference src/models/jat_audiosr_v2.py (differs from V3 only in the
    three norm sites :242,245,361); what train_ddp_v3mod2.py instantiates."""

    _NORM = "ln"


def load_model(checkpoint, device="cuda", cls=None):
    """== load_model (infer_test_v3m2.py:33-94): checkpoint dict (or path) -> eval-mode model on `device`.
    Takes 'config' from the checkpoint when present (defaults :41-53), strips '_orig_mod.' / 'module.'
    prefixes (:64-71), loads with strict=False (:74)."""
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    cfg = dict(checkpoint.get("config", dict(input_channels=1024, cond_channels=1024, patch_len=4, hidden_size=1280,
                                             depth=28, num_q_heads=20, num_kv_heads=4, bottleneck_dim=512,
                                             mlp_ratio=4.0, dropout=0.1, drop_path_rate=0.05)))
    sd = checkpoint["model_state_dict"]
    if any(k.startswith("_orig_mod.") for k in sd):
        sd = {k.replace("_orig_mod.", ""): v for k, v in sd.items()}
    if any(k.startswith("module.") for k in sd):
        sd = {k.replace("module.", ""): v for k, v in sd.items()}
    if cls is None:
        cls = JaT_AudioSR_V3  # what infer_test_v3m2.py:58 instantiates; pass cls=JaT_AudioSR_V2 for v3mod2 (LayerNorm) checkpoints
    model = cls(**cfg)
    missing, unexpected = model.load_state_dict({k: torch.as_tensor(v).float() for k, v in sd.items()}, strict=False)
    model.load_info = dict(missing=list(missing), unexpected=list(unexpected), epoch=checkpoint.get("epoch", 0),
                           global_step=checkpoint.get("global_step", 0))
    return model.to(device).eval()
