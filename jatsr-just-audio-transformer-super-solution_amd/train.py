"""Training step on MI355X — the loop body of reference `train_ddp_v3m2.py:533-622`.

    trainer = Trainer(model, batch_size=28, frames=1378)            # model: jatsr_amd.JaT_AudioSR_V3 on the GPU
    for hr, lr in loader:                                           # raw latents [B, 1024, T]
        stats = trainer.train_step(hr, lr, hr_mean, hr_std, lr_mean, lr_std)

One process per GPU; the only collective is the gradient all-reduce between `jat_trainer_fwd_bwd` and
`jat_trainer_optim` (the reference's DDP hook, train_ddp_v3m2.py:486,610), issued on ONE flat fp32 buffer.  The host
code below owns the hyper-parameters and the RNG draws (torch generators: data, not arithmetic); everything numeric —
normalisation, noise mix, forward, loss, backward, clip, AdamW — runs in the HIP library (include/jat_hip.h).

Dropout (attention probabilities, MLP x2) and DropPath (jat_audiosr_v3.py:38-64,139,175,269-271,300,306) follow the
rates the model was constructed with; their masks come from a counter-based generator keyed by (trainer seed, step,
layer, site, element) inside the kernels — nn.Dropout / drop_path semantics, not torch's Philox stream.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref

import torch

from . import _lib as L

ALIGN = 64   # floats: every tensor starts on a 256-B boundary of the flat buffers


def u_shaped_timestep_sampling(batch_size, device, alpha=0.5, generator=None, u=None):
    """== u_shaped_timestep_sampling (train_ddp_v3m2.py:164-172).  `u` injects the uniform draws (tests)."""
    if u is None:
        u = torch.rand(batch_size, device=device, generator=generator)
    return torch.where(u < 0.5, (2 * u) ** alpha / 2, 1 - ((2 * (1 - u)) ** alpha) / 2)


def get_lr(step, total_steps, warmup_steps, base_lr):
    """== get_lr (train_ddp_v3m2.py:427-432): linear warm-up, then cosine to zero."""
    if step < warmup_steps:
        return base_lr * (step / max(1, warmup_steps))
    progress = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * 0.5 * (1.0 + math.cos(math.pi * progress))


def flat_layout(named_shapes):
    """[(name, shape)] -> ([(name, offset, numel, shape)], total) with ALIGN-float alignment; total % ALIGN == 0."""
    out, off = [], 0
    for name, shape in named_shapes:
        n = 1
        for s in shape:
            n *= int(s)
        out.append((name, off, n, tuple(shape)))
        off += (n + ALIGN - 1) // ALIGN * ALIGN
    return out, off


class GradScaler:
    """The subset of torch.amp.GradScaler the trainer relies on (train_ddp_v3m2.py:435,610-619): a loss scale that is
    halved when a step's gradients are non-finite (the step is skipped) and doubled after `growth_interval` good steps."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.enabled = enabled
        self._good = 0

    def update(self, found_inf: bool):
        if not self.enabled:
            return
        if found_inf:
            self.scale *= self.backoff_factor
            self._good = 0
        else:
            self._good += 1
            if self._good == self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0

    def state_dict(self):
        return dict(scale=self.scale, growth_factor=self.growth_factor, backoff_factor=self.backoff_factor,
                    growth_interval=self.growth_interval, _growth_tracker=self._good)

    def load_state_dict(self, sd):
        self.scale = float(sd["scale"])
        self._good = int(sd.get("_growth_tracker", 0))


def allreduce_mean_(flat, group=None):
    """Sum `flat` over the ranks of `group` in place (reduce-scatter + all-gather, `dist.exchange_sum_`) and return the
    divisor the caller still has to apply (the world size): the division is folded into the optimiser's unscale factor
    instead of a second pass over 3 GB."""
    import torch.distributed as dist
    from .dist import exchange_sum_
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world > 1:
        exchange_sum_(flat, group)
    return world


def reduce_validation_sums(acc, group=None):
    """ONE all-reduce of the 8-slot vector [sum loss, steps, mse, freq, ms, consistency, latent, -] over the ranks, in place:
    what train_ddp_v3mod2.py:1087-1096 does with seven 1-float all-reduces.  Returns (avg_loss, {metric: avg}) of the
    GLOBAL sums / steps (every rank gets the same numbers)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    steps = max(float(acc[1]), 1.0)
    metrics = dict(zip(("mse_loss", "freq_loss", "ms_loss", "consistency_loss", "total_latent_loss"), (acc[2:7] / steps).tolist()))
    return float(acc[0]) / steps, metrics


class Trainer:
    """One rank of the reference training loop.  Hyper-parameter names and defaults are TrainConfig's
    (train_ddp_v3m2.py:55-101)."""

    def __init__(self, model, batch_size, frames, lr=5e-5, weight_decay=0.1, betas=(0.9, 0.999), eps=1e-8,
                 grad_clip=1.0, cfg_dropout_prob=0.1, condition_noise_ratio=0.02, use_adaptive_noise=True,
                 warmup_steps=1000, total_steps=None, use_grad_scaler=True, process_group=None, seed=None,
                 latent_loss_weight=0.0, freq_loss_weight=0.5, ms_loss_weight=0.5, consistency_weight=0.1,
                 low_freq_phase_ratio=0.3, strict_cutoff=0.30, soft_cutoff=0.36, overlap_grad_allreduce=True,
                 distributed=True, amp_dtype=None, loss="mse", charbonnier_eps=1e-6):
        """latent_loss_weight > 0 selects the v3mod2 trainer's loss, MSE + latent perceptual loss
        (train_ddp_v3mod2.py:53-321,362-372,889-896; its TrainConfig uses 0.3 with the other defaults given here, no CFG
        dropout and condition_noise_ratio 0.05); 0 is the MSE-only loss of train_ddp_v3m2.py:585.
        loss: "mse" (F.mse_loss, train_ddp_v3m2.py:585) or "charbonnier" — the V3M2-MOD1 trainer's reconstruction loss
        mean(sqrt((pred - target)^2 + charbonnier_eps)) (train_ddp_v3m2mod1.py:72-101, `use_charbonnier_loss` / `charbonnier_eps`
        :150-151), used for the training step and for validation (:817-819); not combinable with the latent perceptual loss.
        distributed=False: never issue a collective even if a process group exists (a single rank timing a local step).
        amp_dtype: "bf16" (train_ddp_v3m2.py:545) or "fp16" (`torch.amp.autocast('cuda')` of train_ddp_v3mod2.py:854, with
        the dynamic loss scale of :745); must match the operand dtype of the loaded library, which is a process-level
        choice (JAT_OPERAND_DTYPE=fp16 loads libjat_hip_fp16.so).  None: whatever the library is."""
        L.require_gpu()
        have = L.operand_dtype()
        want = {None: have, "bf16": "bf16", "bfloat16": "bf16", "fp16": "fp16", "float16": "fp16"}[amp_dtype]
        if want != have:
            raise L.JatError(f"amp_dtype={want} needs the {want}-operand library: start the process with "
                             f"JAT_OPERAND_DTYPE={want} (loaded: {L.LIB_PATH}, {have})")
        self.amp_dtype = have
        self.model = model
        self.B, self.T = int(batch_size), int(frames)
        self.base_lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.grad_clip = grad_clip
        self.cfg_dropout_prob, self.condition_noise_ratio = cfg_dropout_prob, condition_noise_ratio
        self.use_adaptive_noise = use_adaptive_noise
        self.warmup_steps, self.total_steps = warmup_steps, total_steps
        self.scaler = GradScaler(enabled=use_grad_scaler)
        self.group = process_group
        self.distributed = bool(distributed)
        self.global_step = 0     # every call of optimizer_step (LR schedule, mask seed, checkpoint: train_ddp_v3m2.py:634)
        self.opt_step = 0        # optimiser steps actually taken (AdamW bias correction; skipped when the scaler finds inf)
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise L.JatError("move the model to the GPU first (.to('cuda')); there is no CPU fallback")
        self.device = dev
        prev = model.__dict__.get("_jat_trainer")
        if prev is not None and prev() is not None:
            prev()._detached = True      # the model's parameters move to THIS trainer's flat buffer: the old one must not step
            prev()._release()            # and its C side (15-28 GB of workspace, its second stream and events) goes now, not at GC
            ph = getattr(prev(), "_handle", None)
            if ph is not None:
                ph.trainer = None
        self._detached = False
        import torch.distributed as dist
        rank = dist.get_rank(process_group) if (distributed and dist.is_available() and dist.is_initialized()) else 0
        self.gen = torch.Generator(device=dev)
        # seed=None: decorrelate the ranks' draws of t / noise / CFG mask (each DDP process of the reference has its own
        # generator state); an explicit seed is used as given — pass seed + rank for per-rank streams
        self.gen.manual_seed(seed if seed is not None else 0x5EED0000 + rank)
        # ---- flat fp32 buffers; the model's parameters become views of `params` --------------------------------
        named = [(k, p) for k, p in model.named_parameters()]
        self.layout, total = flat_layout([(k, p.shape) for k, p in named])
        self.params = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        for (k, p), (_, off, n, shape) in zip(named, self.layout):
            view = self.params[off:off + n].view(shape)
            view.copy_(p.data.float())
            p.data = view
        if self._dist_on() and self._world() > 1:
            # DDP broadcasts rank 0's parameters at construction (train_ddp_v3m2.py:512): without it freshly built
            # replicas start from different random weights and never converge to each other
            dist.broadcast(self.params, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)
        h = model._get_handle()          # packs the (now flat-backed) weights
        self._handle = h
        h.trainer = weakref.ref(self)
        model.__dict__["_jat_trainer"] = weakref.ref(self)
        refs = (L.JatTensorRef * len(self.layout))()
        self._keep = []
        for i, (k, off, n, _) in enumerate(self.layout):
            kb = k.encode()
            self._keep.append(kb)
            refs[i] = L.JatTensorRef(kb, self.params.data_ptr() + 4 * off, n)
        self.ptr = C.c_void_p()
        L.check(L.lib().jat_trainer_create(h.ptr, refs, len(self.layout), L.ptr(self.params), L.ptr(self.grads),
                                           L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), total, self.B, self.T,
                                           L.stream_ptr(), C.byref(self.ptr)))
        self._scal = torch.zeros(2, dtype=torch.float32, device=dev)   # loss, scaled grad norm
        self.mask_seed = 0x9E3779B97F4A7C15 if seed is None else int(seed)
        self.set_regularisers([getattr(b, "dropout_rate", 0.0) for b in model.blocks],
                              [getattr(b, "drop_path_rate", 0.0) for b in model.blocks])
        self.latent_loss = dict(latent_weight=float(latent_loss_weight), freq_weight=float(freq_loss_weight),
                                ms_weight=float(ms_loss_weight), consistency_weight=float(consistency_weight),
                                low_freq_phase_ratio=float(low_freq_phase_ratio), strict_cutoff=float(strict_cutoff),
                                soft_cutoff=float(soft_cutoff))
        if loss not in ("mse", "charbonnier"):
            raise ValueError(f"loss must be 'mse' or 'charbonnier', got {loss!r}")
        if loss == "charbonnier" and float(latent_loss_weight) != 0.0:
            raise ValueError("the latent perceptual loss is defined on top of the MSE loss (train_ddp_v3mod2.py:889-896); "
                             "the Charbonnier trainer (train_ddp_v3m2mod1.py) has no latent term")
        self.loss, self.charbonnier_eps = loss, float(charbonnier_eps)
        L.check(L.lib().jat_trainer_set_latent_loss(self.ptr, *self.latent_loss.values()))
        L.check(L.lib().jat_trainer_set_charbonnier(self.ptr, self.charbonnier_eps if loss == "charbonnier" else 0.0))
        self._terms = torch.zeros(6, dtype=torch.float32, device=dev)
        # gradient all-reduce overlapped with the backward: one async all-reduce per parameter slice as soon as its
        # last gradient kernel is enqueued (jat_trainer_set_grad_hook), on a side stream ordered by an event
        self.overlap = overlap_grad_allreduce      # True: when world_size > 1; "force": also with one rank (tests)
        self._pending, self._covered = [], 0
        self._comm_stream = torch.cuda.Stream(device=dev)
        self._hook = L.GRAD_HOOK(self._on_grads_ready)          # keep the ctypes thunk alive
        L.check(L.lib().jat_trainer_set_grad_hook(self.ptr, C.cast(self._hook, C.c_void_p), None))

    def _check_attached(self):
        if self._detached:
            raise L.JatError("this Trainer was superseded: a newer Trainer owns the model's parameters (they are views of "
                             "the newer trainer's flat buffer)")

    def _dist_on(self):
        import torch.distributed as dist
        return self.distributed and dist.is_available() and dist.is_initialized()

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if self._dist_on() else 1

    def _on_grads_ready(self, off, n, _user):
        import torch.distributed as dist
        if not self.overlap or not self._dist_on():
            return
        if self._world() == 1 and self.overlap != "force":
            return
        from .dist import exchange_sum_
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(ev)
            if self._world() == 1:     # overlap == "force": exercise the hook machinery with a 1-rank collective
                self._pending.append(dist.all_reduce(self.grads[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:                      # reduce-scatter + all-gather of this slice over all xGMI links
                self._pending.extend(exchange_sum_(self.grads[off:off + n], self.group, async_op=True))
        self._covered += n

    def loss_terms(self):
        """{total, mse, freq, ms, consistency, latent} of the latest step (the trainer's `latent_loss_dict`,
        train_ddp_v3mod2.py:313-318); MSE-only trainers return {total}."""
        if self.latent_loss["latent_weight"] == 0.0:
            return dict(total=float(self._scal[0]))
        L.check(L.lib().jat_trainer_loss_terms(self.ptr, L.ptr(self._terms), L.stream_ptr()))
        return dict(zip(("total", "mse", "freq", "ms", "consistency", "latent"), self._terms.tolist()))

    def set_regularisers(self, dropout, drop_path):
        """Per-layer nn.Dropout p and DropPath rate (defaults: what the model was constructed with,
        jat_audiosr_v3.py:372-377)."""
        n = len(self.model.blocks)
        if len(dropout) != n or len(drop_path) != n:
            raise ValueError(f"need {n} per-layer rates")
        self.dropout, self.drop_path = [float(x) for x in dropout], [float(x) for x in drop_path]
        L.check(L.lib().jat_trainer_set_regularisers(self.ptr, (C.c_float * n)(*self.dropout), (C.c_float * n)(*self.drop_path)))

    def step_seed(self, step=None):
        """64-bit mask seed of a step: splitmix64 of (trainer seed, step index, rank)."""
        import torch.distributed as dist
        rank = dist.get_rank(self.group) if self._dist_on() else 0
        x = (self.mask_seed + 0x9E3779B97F4A7C15 * ((self.global_step if step is None else step) * 4096 + rank + 1)) & (2 ** 64 - 1)
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        return x ^ (x >> 31)

    def _release(self):
        """Destroy the C-side trainer (workspace, streams, events).  The flat torch buffers stay with their owners."""
        if getattr(self, "ptr", None):
            torch.cuda.synchronize(self.device)     # nothing of this trainer may still be in flight on any stream
            L.lib().jat_trainer_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # -- views -------------------------------------------------------------------------------------------------
    def grad(self, name):
        """Gradient of the named parameter as a view of the flat buffer (after `forward_backward`)."""
        for k, off, n, shape in self.layout:
            if k == name:
                return self.grads[off:off + n].view(shape)
        raise KeyError(name)

    def workspace_bytes(self):
        out = C.c_size_t()
        L.check(L.lib().jat_trainer_workspace_bytes(self.ptr, C.byref(out)))
        return out.value

    # -- the step, in the reference's order -------------------------------------------------------------------------
    def prepare(self, hr_norm, lr_norm, noise=None, cond_noise=None, cfg_mask=None, t=None):
        """train_ddp_v3m2.py:548-579 on normalised latents.  Draws whatever is not injected.  Returns (z_t, t, cond)."""
        B = self.B
        if t is None:
            t = u_shaped_timestep_sampling(B, self.device, generator=self.gen)
        if noise is None:
            noise = torch.randn(hr_norm.shape, device=self.device, generator=self.gen)
        if cond_noise is None and self.condition_noise_ratio > 0:
            cond_noise = torch.randn(lr_norm.shape, device=self.device, generator=self.gen)
        if cfg_mask is None:
            cfg_mask = torch.rand(B, device=self.device, generator=self.gen) < self.cfg_dropout_prob
        keep = (~cfg_mask.to(self.device).bool()).float().contiguous()
        cond = lr_norm.contiguous().clone()
        z_t = torch.empty_like(hr_norm)
        t = t.to(self.device, torch.float32).contiguous()
        L.check(L.lib().jat_trainer_prepare(self.ptr, L.ptr(hr_norm.contiguous()), L.ptr(cond), L.ptr(noise.contiguous()),
                                            L.ptr(cond_noise.contiguous()) if cond_noise is not None else None,
                                            float(self.condition_noise_ratio if cond_noise is not None else 0.0),
                                            int(self.use_adaptive_noise), L.ptr(keep), L.ptr(t), L.ptr(z_t), L.stream_ptr()))
        return z_t, t, cond

    def forward_backward(self, z_t, t, cond, target, want_pred=False, mask_seed=None, cond_clean=None):
        """pred = model(z_t, t, cond); loss(pred, target[, cond_clean]); backward -> self.grads (scaled by scaler.scale).
        mask_seed: 64-bit seed of this step's Dropout / DropPath masks (default: `step_seed()`).
        cond_clean: the normalised LR latent before the condition noise (`lr_norm_original`, train_ddp_v3mod2.py:861),
        needed by the consistency term of the latent perceptual loss."""
        for x in (z_t, cond, target) + ((cond_clean,) if cond_clean is not None else ()):
            if tuple(x.shape) != (self.B, self.model.input_channels, self.T) or x.dtype != torch.float32 or not x.is_cuda:
                raise ValueError(f"expected fp32 CUDA [{self.B}, {self.model.input_channels}, {self.T}], got "
                                 f"{tuple(x.shape)} {x.dtype} on {x.device}")
        self._check_attached()
        self.model._get_handle()     # parameters overwritten through PyTorch (load_state_dict)? re-pack, incl. this trainer's copies
        pred = torch.empty_like(z_t) if want_pred else None
        self._pending, self._covered = [], 0
        L.check(L.lib().jat_trainer_fwd_bwd(self.ptr, L.ptr(z_t.contiguous()), L.ptr(t.contiguous()), L.ptr(cond.contiguous()),
                                            L.ptr(target.contiguous()),
                                            L.ptr(cond_clean.contiguous()) if cond_clean is not None else None,
                                            float(self.scaler.scale),
                                            C.c_uint64(self.step_seed() if mask_seed is None else int(mask_seed)),
                                            L.ptr(self._scal),
                                            L.ptr(pred) if want_pred else None, L.stream_ptr()))
        return pred

    def optimizer_step(self, lr=None):
        """All-reduce, unscale, clip_grad_norm_(grad_clip), AdamW, re-pack.  Returns (loss, grad_norm) as floats —
        the one host synchronisation of the step, like the reference's `.item()` calls (train_ddp_v3m2.py:615,622)."""
        self._check_attached()
        if self._pending:          # slices were reduced under the backward: the step's stream waits for the last of them
            for w in self._pending:
                w.wait()
            assert self._covered == self.grads.numel(), "gradient hooks did not tile the flat buffer"
            self._pending, self._covered = [], 0
            world = self._world()
        else:
            world = allreduce_mean_(self.grads, self.group) if self._dist_on() else 1
        if lr is None:
            lr = get_lr(self.global_step, self.total_steps, self.warmup_steps, self.base_lr) if self.total_steps else self.base_lr
        scale = self.scaler.scale * world
        L.check(L.lib().jat_trainer_optim(self.ptr, float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                          float(self.weight_decay), float(self.grad_clip or 0.0), float(scale),
                                          self.opt_step + 1, C.c_void_p(self._scal.data_ptr() + 4), L.stream_ptr()))
        loss, gnorm = self._scal.tolist()
        gnorm /= scale
        found_inf = not math.isfinite(gnorm)    # the same on every rank: the norm is taken over the all-reduced gradients
        self.scaler.update(found_inf)
        self.global_step += 1                   # counts batches, skipped or not (train_ddp_v3m2.py:634)
        if not found_inf:
            self.opt_step += 1
            self._handle.epoch += 1      # the weights changed under the model: cached samplers (mod tables, graphs) are stale
        self.last_lr = lr
        return loss, gnorm

    def train_step(self, hr, lr, hr_mean, hr_std, lr_mean, lr_std):
        """Raw latents in, one optimisation step (train_ddp_v3m2.py:533-622).  Returns dict(loss, grad_norm, lr)."""
        from .sampler import channel_affine
        hr_norm = channel_affine(hr.to(self.device, torch.float32), hr_mean, hr_std)
        lr_norm = channel_affine(lr.to(self.device, torch.float32), lr_mean, lr_std)
        z_t, t, cond = self.prepare(hr_norm, lr_norm)
        self.forward_backward(z_t, t, cond, hr_norm, cond_clean=lr_norm)
        loss, gnorm = self.optimizer_step()
        return dict(loss=loss, grad_norm=gnorm, lr=self.last_lr, step=self.global_step)

    # -- validation (train_ddp_v3mod2.py:1026-1118 / train_ddp_v3m2.py:695-745) -------------------------------------------
    @torch.no_grad()
    def validate(self, batches, hr_mean, hr_std, lr_mean, lr_std, t=None, noise=None):
        """Eval-mode loss over an iterable of (hr, lr) raw-latent batches: uniform t, no condition noise, no CFG dropout,
        no Dropout / DropPath; the same loss as the training step.  Returns (avg_loss, loss_std, metrics) — the
        reference's triple; the sums of all ranks are combined by ONE all-reduce of an 8-float vector (the reference
        issues seven 1-float all-reduces, :1087-1096).  `t` / `noise`: optional per-batch lists (tests)."""
        from .sampler import channel_affine
        acc = torch.zeros(8, dtype=torch.float64, device=self.device)   # loss, steps, mse, freq, ms, cons, latent, -
        losses = []
        ll = self.latent_loss
        out6 = torch.zeros(6, dtype=torch.float32, device=self.device)
        for i, (hr, lr) in enumerate(batches):
            hr_norm = channel_affine(hr.to(self.device, torch.float32), hr_mean, hr_std)
            lr_norm = channel_affine(lr.to(self.device, torch.float32), lr_mean, lr_std)
            Bv, Cv, Tv = hr_norm.shape
            tt = t[i].to(self.device, torch.float32) if t is not None else torch.rand(Bv, device=self.device, generator=self.gen)
            nz = noise[i].to(self.device) if noise is not None else torch.randn(hr_norm.shape, device=self.device, generator=self.gen)
            tv = tt.view(-1, 1, 1)
            z_t = tv * hr_norm + (1 - tv) * nz                          # plumbing-sized elementwise op, as in the reference
            pred = self.model(z_t.contiguous(), tt.contiguous(), lr_norm)
            rows = Bv * Cv
            scratch = torch.empty_like(pred)
            if self.loss == "charbonnier":      # train_ddp_v3m2mod1.py:817-819: validation uses the training loss
                work = torch.empty(4104, dtype=torch.uint8, device=self.device)
                L.check(L.lib().jat_k_recon_loss(L.ptr(pred), L.ptr(hr_norm), L.ptr(scratch), L.ptr(out6), pred.numel(),
                                                 self.charbonnier_eps, 1.0, L.ptr(work), work.numel(), L.stream_ptr()))
                acc[0] += out6[0].double(); acc[1] += 1
                losses.append(float(out6[0]))
                continue
            work = torch.empty((Tv * 8 + 255) // 256 * 256 + rows * 32, dtype=torch.uint8, device=self.device)
            L.check(L.lib().jat_k_latent_loss(L.ptr(pred), L.ptr(hr_norm), L.ptr(lr_norm), L.ptr(scratch), L.ptr(out6), rows, Tv,
                                              ll["latent_weight"], ll["freq_weight"], ll["ms_weight"], ll["consistency_weight"],
                                              ll["low_freq_phase_ratio"], ll["strict_cutoff"], ll["soft_cutoff"], 1.0,
                                              L.ptr(work), work.numel(), L.stream_ptr()))
            o = out6.double()
            acc[0] += o[0]; acc[1] += 1; acc[2:7] += o[1:6]
            losses.append(float(o[0]))
        if self._dist_on():
            avg, metrics = reduce_validation_sums(acc, self.group)
        else:
            steps = max(float(acc[1]), 1.0)
            avg = float(acc[0]) / steps
            metrics = dict(zip(("mse_loss", "freq_loss", "ms_loss", "consistency_loss", "total_latent_loss"),
                               (acc[2:7] / steps).tolist()))
        std = float(torch.tensor(losses).std()) if len(losses) > 1 else 0.0
        if ll["latent_weight"] == 0.0:
            metrics = {}
        return avg, std, metrics

    # -- checkpoint egress / ingest in the reference's layout (train_ddp_v3m2.py:747-770, 443-500) ----------------------
    def optimizer_state_dict(self):
        """torch.optim.AdamW.state_dict() layout, so that the reference trainer can resume from it."""
        state = {}
        for i, (k, off, n, shape) in enumerate(self.layout):
            state[i] = dict(step=torch.tensor(float(self.opt_step)),
                            exp_avg=self.exp_avg[off:off + n].view(shape).clone(),
                            exp_avg_sq=self.exp_avg_sq[off:off + n].view(shape).clone())
        group = dict(lr=getattr(self, "last_lr", self.base_lr), betas=tuple(self.betas), eps=self.eps,
                     weight_decay=self.weight_decay, amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, params=list(range(len(self.layout))))
        return dict(state=state if self.opt_step > 0 else {}, param_groups=[group])

    def load_optimizer_state_dict(self, sd):
        for i, (k, off, n, shape) in enumerate(self.layout):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.exp_avg[off:off + n].view(shape).copy_(st["exp_avg"].to(self.device, torch.float32))
            self.exp_avg_sq[off:off + n].view(shape).copy_(st["exp_avg_sq"].to(self.device, torch.float32))
            self.opt_step = int(float(st["step"]))       # AdamW's own counter: bias correction resumes where it stopped

    def save_checkpoint(self, path, epoch=0, best_val_loss=float("inf")):
        ck = dict(epoch=epoch, global_step=self.global_step, best_val_loss=best_val_loss,
                  model_state_dict={k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                  optimizer_state_dict=self.optimizer_state_dict(), scaler_state_dict=self.scaler.state_dict(),
                  config=dict(self.model.config(), dropout=max(self.dropout), drop_path_rate=max(self.drop_path)))
        torch.save(ck, path)
        return ck

    def load_checkpoint(self, checkpoint):
        """Resume (train_ddp_v3m2.py:443-500): model weights (prefixes stripped, strict=False), AdamW moments, step
        counter and loss scale from a checkpoint dict or path in the reference's layout."""
        if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
            checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
        sd = checkpoint["model_state_dict"]
        sd = {k.replace("_orig_mod.", "").replace("module.", ""): v for k, v in sd.items()}
        own = dict(self.model.named_parameters())
        for k, v in sd.items():
            if k in own:
                own[k].data.copy_(torch.as_tensor(v).to(self.device, torch.float32))   # writes through to the flat buffer
        if checkpoint.get("optimizer_state_dict"):
            self.load_optimizer_state_dict(checkpoint["optimizer_state_dict"])
        if checkpoint.get("scaler_state_dict") and self.scaler.enabled:
            self.scaler.load_state_dict(checkpoint["scaler_state_dict"])
        self.global_step = int(checkpoint.get("global_step", self.opt_step))
        self._weights_replaced()
        return checkpoint.get("epoch", 0)

    def _weights_replaced(self):
        """The fp32 master weights were overwritten from outside an optimiser step (checkpoint, load_state_dict): rebuild
        every operand copy (bf16, transposed) and make dependants stale.  Synchronous: a sampler created next builds its
        tables on a private stream and must see the finished copies."""
        if self._detached or not self.ptr:      # superseded: the newer Trainer owns the weights and re-packs them itself
            return
        L.check(L.lib().jat_trainer_repack(self.ptr, L.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        self._handle.epoch += 1
