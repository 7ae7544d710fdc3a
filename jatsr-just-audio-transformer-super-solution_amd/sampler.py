"""CFG flow-matching sampler and chunk driver — host-side mirror of reference infer_test_v3m2.py.

`flow_matching_sample(model, lr_latent, num_steps, cfg_scale, device, verbose)` keeps the reference
signature (infer_test_v3m2.py:107-185) and adds an optional `z0` (initial noise; the reference draws it
with torch.randn at :133).  The whole loop — CFG double batch, 28-block forward, CFG combine, Euler update —
runs as ONE hipGraph replay inside libjat_hip.so (`jat_sampler_run`); the schedule `linspace(0,1,steps+1)`
and the `t < 0.999` branch (:173) are evaluated on the host, so no device scalar is ever read back.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class Sampler:
    """A captured sampler for one (model, B, T, steps, cfg_scale) bucket: `jat_sampler_create` (include/jat_hip.h)."""

    def __init__(self, model, B, T, num_steps=50, cfg_scale=1.0):
        self.model = model
        self.B, self.T, self.steps, self.cfg_scale = int(B), int(T), int(num_steps), float(cfg_scale)
        self.ptr = C.c_void_p()
        self._build()

    def _build(self):
        """(Re)create the C-side sampler: modulation table, per-step folded weights and the captured graph all depend on
        the model's weights as they are NOW."""
        self._destroy()
        h = self.model._get_handle()
        self._handle = h
        self._version = (h.version, getattr(h, "epoch", 0))   # epoch: bumped by jatsr_amd.train after every weight update
        self._lengths = None                                  # per-row valid frames currently set in the C-side sampler
        L.check(L.lib().jat_sampler_create(h.ptr, self.B, self.T, self.steps, self.cfg_scale, C.byref(self.ptr)))

    def _destroy(self):
        if self.ptr:
            L.lib().jat_sampler_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def info(self):
        """{'folded': per-step folded weights in use, 'fused_attn': fused QKV+RoPE+attention kernel, 'fold_bytes': table size}
        (`jat_sampler_info`): what the captured graph of this bucket runs."""
        f, a, n = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        L.check(L.lib().jat_sampler_info(self.ptr, C.byref(f), C.byref(a), C.byref(n)))
        return {"folded": bool(f.value), "fused_attn": bool(a.value), "fold_bytes": int(n.value)}

    def run(self, lr_latent, z0, use_graph=True, lengths=None):
        """lengths: optional valid frame count per batch row (rows shorter than the bucket's T, zero-padded by the caller:
        the last chunk of a file batched with the full-length ones).  Their valid frames equal a stand-alone run."""
        lr_latent = lr_latent.detach().to(torch.float32).contiguous()
        z0 = z0.detach().to(torch.float32).contiguous()
        if tuple(lr_latent.shape) != (self.B, self.model.input_channels, self.T) or z0.shape != lr_latent.shape:
            raise ValueError(f"sampler bucket is [B={self.B}, C={self.model.input_channels}, T={self.T}], got "
                             f"lr {tuple(lr_latent.shape)} z0 {tuple(z0.shape)}")
        # a sampler held across Trainer.optimizer_step / load_checkpoint / load_state_dict would mix the OLD modulation table
        # and folded weights with the NEW packed weights: rebuild it instead
        h = self.model._get_handle()
        if (h.version, getattr(h, "epoch", 0)) != self._version:
            self._build()
        if lengths is None and self._lengths is not None:
            lengths = [self.T] * self.B                     # back to full-length rows
        if lengths is not None:
            lengths = [int(v) for v in lengths]
            if lengths != self._lengths:
                arr = (C.c_int32 * self.B)(*lengths)
                L.check(L.lib().jat_sampler_set_lengths(self.ptr, arr, self.B, L.stream_ptr()))
                self._lengths = lengths if any(v != self.T for v in lengths) else None
        out = torch.empty_like(z0)
        L.check(L.lib().jat_sampler_run(self.ptr, L.ptr(lr_latent), L.ptr(z0), L.ptr(out), 1 if use_graph else 0,
                                        L.stream_ptr()))
        return out


def _cached_sampler(model, B, T, num_steps, cfg_scale):
    cache = model.__dict__.setdefault("_jat_samplers", {})
    key = (B, T, num_steps, float(cfg_scale))
    s = cache.get(key)
    h = model._get_handle()  # repacks if the weights changed
    if s is None or s._version != (h.version, getattr(h, "epoch", 0)):
        s = Sampler(model, B, T, num_steps, cfg_scale)
        cache[key] = s
    return s


@torch.no_grad()
def flow_matching_sample(model, lr_latent, num_steps=50, cfg_scale=1.0, device="cuda", verbose=True, z0=None,
                         use_graph=True, lengths=None):
    """Flow-matching Euler sampling with CFG (x-prediction), reference infer_test_v3m2.py:107-185.

    lr_latent: [B, C, T] normalised LR latent.  Returns the generated [B, C, T] latent.
    """
    L.require_gpu()
    lr_latent = lr_latent.to(device)
    B, Cc, T = lr_latent.shape
    if z0 is None:
        z0 = torch.randn(B, Cc, T, device=lr_latent.device)      # :133
    if verbose:
        print(f"  Flow Matching sampling ({num_steps} steps, CFG scale={cfg_scale}) [hipGraph={bool(use_graph)}]")
    s = _cached_sampler(model, B, T, num_steps, cfg_scale)
    return s.run(lr_latent, z0.to(lr_latent.device), use_graph=use_graph, lengths=lengths)


def crossfade_chunks(chunks, overlap_frames):
    """Linear crossfade concatenation of [1, C, T_i] chunks — reference infer_test_v3m2.py:188-233."""
    if len(chunks) == 0:
        return None
    if len(chunks) == 1:
        return chunks[0]
    L.require_gpu()
    result = chunks[0].contiguous()
    for cur in chunks[1:]:
        cur = cur.contiguous()
        ov = overlap_frames if (overlap_frames > 0 and result.shape[-1] >= overlap_frames) else 0   # :209, :229-231
        rows = result.shape[0] * result.shape[1]
        Tp, Tc = result.shape[-1], cur.shape[-1]
        out = torch.empty(result.shape[0], result.shape[1], Tp + Tc - ov, dtype=torch.float32, device=result.device)
        L.check(L.lib().jat_crossfade_pair(L.ptr(result), Tp, L.ptr(cur), Tc, ov, L.ptr(out), rows, L.stream_ptr()))
        result = out
    return result


def chunk_plan(total_frames, chunk_frames=1378, overlap_frames=172):
    """(start, end) of every chunk — reference infer_test_v3m2.py:340-361,370-372 (16 s chunks, 2 s overlap)."""
    stride = chunk_frames - overlap_frames
    num = (total_frames - overlap_frames + stride - 1) // stride
    return [(i * stride, min(i * stride + chunk_frames, total_frames)) for i in range(num)]


def channel_affine(x, mean, std, inverse=False):
    """(x - mean_c)/std_c or x*std_c + mean_c per channel — reference infer_test_v3m2.py:381-382,394."""
    L.require_gpu()
    x = x.contiguous()
    B, Cc, T = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().jat_channel_affine(L.ptr(x), L.ptr(mean.contiguous().view(-1)), L.ptr(std.contiguous().view(-1)),
                                       L.ptr(out), B, Cc, T, 1 if inverse else 0, L.stream_ptr()))
    return out


def chunk_groups(lens, pad_short_chunks=True):
    """Which chunks share a sampler launch: {bucket length T: [chunk indices]}.  One bucket per chunk length ... except that
    SHORTER chunks ride along with the longest ones, zero-padded, their padded keys masked in attention and their padded frames
    read as zeros (`Sampler.run(lengths=...)`): one launch for the whole file instead of a second, latency-bound one for a few
    hundred frames.  A bucket of exactly 128 tokens runs the fused QKV+attention kernel, which has no key mask
    (jat_sampler_set_lengths refuses it): there every length keeps its own bucket.  The ONE place that knows this rule
    (sample_long, bench.py --mode long and dist.sample_long_sharded all come here)."""
    Tmax = max(lens)
    if pad_short_chunks and len(set(lens)) > 1 and (Tmax + 3) // 4 != 128:
        return {Tmax: list(range(len(lens)))}
    groups = {}
    for i, n in enumerate(lens):
        groups.setdefault(n, []).append(i)
    return groups


@torch.no_grad()
def sample_long(model, lr_latent, hr_mean, hr_std, lr_mean, lr_std, num_steps=50, cfg_scale=1.0,
                chunk_frames=1378, overlap_frames=172, noise=None, pad_short_chunks=True):
    """Chunked long-sequence inference == the chunk loop of infer_test_v3m2.py:340-404, with the chunks of a file
    BATCHED into one sampler launch instead of the reference's serial B=1 loop (pad_short_chunks=False: one launch
    per distinct chunk length, the round-1 behaviour).

    lr_latent: [C, T_total] un-normalised latent.  noise: optional list of per-chunk z0 tensors [1,C,T_i].
    Returns [1, C, T_total] de-normalised generated latent.
    """
    Cc, total = lr_latent.shape
    plan = chunk_plan(total, chunk_frames, overlap_frames)
    lr = lr_latent.unsqueeze(0)
    lens = [b - a for a, b in plan]
    outs = [None] * len(plan)
    groups = chunk_groups(lens, pad_short_chunks)
    for length, idxs in groups.items():
        rows, noise_rows = [], []
        for i in idxs:
            c = channel_affine(lr[:, :, plan[i][0]:plan[i][1]], lr_mean, lr_std)
            z = None if noise is None else noise[i]
            if lens[i] < length:                       # zero-pad to the bucket's T (containers only: no arithmetic here)
                cp = torch.zeros(1, Cc, length, dtype=c.dtype, device=c.device)
                cp[:, :, :lens[i]] = c
                c = cp
                if z is not None:
                    zp = torch.zeros(1, Cc, length, dtype=z.dtype, device=z.device)
                    zp[:, :, :lens[i]] = z
                    z = zp
            rows.append(c)
            noise_rows.append(z)
        batch = torch.cat(rows, 0)
        if noise is None:
            z0 = torch.randn(batch.shape, device=batch.device)
            for j, i in enumerate(idxs):
                z0[j, :, lens[i]:] = 0
        else:
            z0 = torch.cat(noise_rows, 0)
        row_lens = [lens[i] for i in idxs]
        gen = flow_matching_sample(model, batch, num_steps, cfg_scale, device=batch.device, verbose=False, z0=z0,
                                   lengths=row_lens if any(v != length for v in row_lens) else None)
        gen = channel_affine(gen, hr_mean, hr_std, inverse=True)
        for j, i in enumerate(idxs):
            outs[i] = gen[j:j + 1, :, :lens[i]].contiguous()
    return crossfade_chunks(outs, overlap_frames)
