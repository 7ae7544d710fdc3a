"""CPU side of the training step: the numpy oracle (hand-derived backward) is PINNED against the
reference-autograd goldens, and the host logic of `jatsr_amd.train` (schedule, t sampling, flat layout, loss-scale
bookkeeping, gradient all-reduce over gloo with world_size 2) is checked without a GPU."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch

import jatsr_amd.recipe as recipe
from helpers import load_golden, rel_l2
from jatsr_amd import train as T
from oracle import jat_oracle_train as OT


def gsub(a, meta):
    s = meta["strides"]
    a = np.asarray(a)
    if a.size <= meta["full_limit"] or a.ndim != 2:
        return a if a.size <= meta["full_limit"] else a.reshape(-1)[::(meta.get("stride1d") or s[0] * s[1])]
    return a[::s[0], ::s[1]]


def oracle_step(meta):
    cfg = recipe.CONFIGS[meta["cfg"]]
    C, B, Tn, salt = cfg["input_channels"], meta["B"], meta["T"], meta["salt"]
    sd = recipe.make_state_dict(cfg, meta["norm"], salt)
    hr = recipe.gaussian("train_hr", (B, C, Tn), salt + 300).astype(np.float64)
    lr = recipe.gaussian("train_lr", (B, C, Tn), salt + 301).astype(np.float64)
    noise = recipe.gaussian("train_noise", (B, C, Tn), salt + 302).astype(np.float64)
    t = np.asarray(meta["t"], np.float32).astype(np.float64)
    keep = (~np.asarray(meta["mask"], bool)).astype(np.float64).reshape(B, 1, 1)
    tv = t.reshape(B, 1, 1)
    z_t = tv * hr + (1 - tv) * noise                     # train_ddp_v3m2.py:577-579
    orc = OT.TrainOracle(cfg, sd, meta["norm"])
    charb = meta.get("charbonnier_eps") if meta.get("loss") == "charbonnier" else None
    loss, grads, pred = orc.loss_and_grads(z_t, t, lr * keep, hr, charbonnier_eps=charb)
    return orc, loss, grads, pred


@pytest.mark.parametrize("name", ["train_charbonnier_T24", "train_charbonnier_T1378"])
def test_charbonnier_oracle_matches_reference_function(name):
    """oracle charbonnier_loss == the reference's own function (train_ddp_v3m2mod1.py:72-101, AST-extracted, autograd):
    value and d/d pred, including elements with |pred - target| ~ sqrt(eps) and exactly 0."""
    z, meta = load_golden(name)
    B, C, Tn, salt = meta["B"], meta["C"], meta["T"], meta["salt"]
    pred = recipe.gaussian("charb_pred", (B, C, Tn), salt + 500)
    target = recipe.gaussian("charb_target", (B, C, Tn), salt + 501)
    near = recipe.gaussian("charb_near", (B, C, Tn), salt + 502)
    ft, fp, fn = target.reshape(-1), pred.reshape(-1), near.reshape(-1)
    ft[::3] = fp[::3] + 2e-3 * fn[::3]
    ft[::9] = fp[::9]
    loss, dpred = OT.charbonnier_loss(pred, target, meta["eps"])
    assert abs(loss - float(z["loss64"])) <= 1e-12
    assert np.abs(dpred - z["dpred64"]).max() <= 1e-15
    assert np.all(dpred.reshape(-1)[::9] == 0.0)                       # d == 0: zero gradient, loss sqrt(eps) per element
    assert abs(float(z["loss32"]) - loss) <= 1e-6                       # the reference's own fp32 run


@pytest.mark.parametrize("name", ["train_micro_T24", "train_micro_T22_pad", "train_micro_ln_T24", "train_tiny_T128",
                                  "train_tiny_T1378", "train_micro_charbonnier_T24", "train_tiny_charbonnier_T128"])
def test_train_oracle_matches_reference_autograd(name):
    z, meta = load_golden(name)
    orc, loss, grads, pred = oracle_step(meta)
    assert abs(loss - float(z["loss64"])) <= 1e-9 * float(z["loss64"])
    assert abs(np.linalg.norm(pred) - float(z["pred_l2"])) <= 1e-9 * float(z["pred_l2"])
    assert sorted(grads) == sorted(meta["names"])          # every trainable tensor of the reference, nothing else
    for k in meta["names"]:
        g = grads[k]
        ref_l2 = float(z["gl2_" + k])
        assert abs(np.linalg.norm(g) - ref_l2) <= 1e-8 * max(ref_l2, 1e-30), k
        # the fixture stores fp32 values of the fp64 run
        assert rel_l2(gsub(g, meta), z["g_" + k]) <= 2e-7, k
    params = {k: orc.sd[k] for k in meta["names"]}
    gnorm, new, _ = OT.TrainOracle.clip_and_adamw(params, grads, meta["lr"], meta["wd"], meta["clip"])
    assert abs(gnorm - float(z["gnorm64"])) <= 1e-9 * gnorm
    for k in meta["names"]:
        d = new[k] - params[k]
        assert abs(np.linalg.norm(d) - float(z["dl2_" + k])) <= 1e-6 * float(z["dl2_" + k]) + 1e-12, k
        assert np.abs(gsub(d, meta) - z["d_" + k]).max() <= 1e-3 * meta["lr"], k


@pytest.mark.parametrize("name", ["train_micro_drop_T24", "train_tiny_drop_T128"])
def test_train_oracle_dropout_semantics_match_reference(name):
    """Train-mode Dropout / DropPath: the reference ran with the counter-based masks injected at its own random calls
    (oracle/gen_golden_train.py `dropout_case`); the oracle applies the same masks at the cited lines."""
    z, meta = load_golden(name)
    cfg = recipe.CONFIGS[meta["cfg"]]
    C, B, Tn, salt = cfg["input_channels"], meta["B"], meta["T"], meta["salt"]
    sd = recipe.make_state_dict(cfg, "rms", salt)
    hr = recipe.gaussian("train_hr", (B, C, Tn), salt + 300).astype(np.float64)
    lr = recipe.gaussian("train_lr", (B, C, Tn), salt + 301).astype(np.float64)
    noise = recipe.gaussian("train_noise", (B, C, Tn), salt + 302).astype(np.float64)
    t = np.asarray(meta["t"], np.float32).astype(np.float64)
    tv = t.reshape(B, 1, 1)
    plan = OT.DropPlan(meta["seed"], [meta["dropout"]] * cfg["depth"], meta["drop_path"])
    # the masks are not degenerate: something is dropped at every kind of site
    assert (plan.mult(1, 2, (B, meta["N"], meta["mlp"])) == 0).mean() == pytest.approx(meta["dropout"], abs=0.02)
    assert any((plan.mult(l, k, (B,)) == 0).any() for l in range(cfg["depth"]) for k in (1, 4))
    loss, grads, pred = OT.TrainOracle(cfg, sd, "rms").loss_and_grads(tv * hr + (1 - tv) * noise, t, lr, hr, plan)
    assert abs(loss - float(z["loss64"])) <= 1e-9 * float(z["loss64"])
    for k in meta["names"]:
        ref_l2 = float(z["gl2_" + k])
        assert abs(np.linalg.norm(grads[k]) - ref_l2) <= 1e-7 * max(ref_l2, 1e-30), k
        assert rel_l2(gsub(grads[k], meta), z["g_" + k]) <= 2e-7, k


def test_u_shaped_sampling_and_schedule():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_misc.npz"))
    t = T.u_shaped_timestep_sampling(64, "cpu", u=torch.from_numpy(z["u"]))
    assert np.array_equal(t.numpy(), z["t_ushape"])                     # same torch ops as the reference: bit-exact
    assert np.allclose(OT.u_shaped_timestep_sampling(z["u"]), z["t_ushape"], atol=1e-7)
    assert t.min() >= 0 and t.max() <= 1
    # get_lr: train_ddp_v3m2.py:427-432
    assert T.get_lr(0, 10000, 1000, 5e-5) == 0.0
    assert T.get_lr(500, 10000, 1000, 5e-5) == pytest.approx(2.5e-5)
    assert T.get_lr(1000, 10000, 1000, 5e-5) == pytest.approx(5e-5)
    assert T.get_lr(5500, 10000, 1000, 5e-5) == pytest.approx(2.5e-5)
    assert T.get_lr(10000, 10000, 1000, 5e-5) == pytest.approx(0.0, abs=1e-12)


def test_flat_layout_and_grad_scaler():
    shapes = recipe.model_param_shapes(recipe.CONFIGS["micro"], "rms")
    lay, total = T.flat_layout([(k, s) for k, s in shapes.items() if ".rope." not in k])
    assert total % T.ALIGN == 0
    end = 0
    for name, off, n, shape in lay:
        assert off % T.ALIGN == 0 and off >= end and n == int(np.prod(shape))
        end = off + n
    assert end <= total
    s = T.GradScaler(init_scale=1024.0, growth_interval=3)
    s.update(True)
    assert s.scale == 512.0
    for _ in range(3):
        s.update(False)
    assert s.scale == 1024.0
    s2 = T.GradScaler()
    s2.load_state_dict(s.state_dict())
    assert s2.scale == s.scale
    assert T.GradScaler(enabled=False).scale == 1.0


def test_trainer_needs_the_gpu_and_reads_the_model_rates():
    import jatsr_amd
    import jatsr_amd._lib as L
    m = jatsr_amd.JaT_AudioSR_V3(**recipe.CONFIGS["micro"], dropout=0.0, drop_path_rate=0.0)
    if not torch.cuda.is_available():
        with pytest.raises(L.JatError):
            T.Trainer(m, batch_size=2, frames=24)        # no CPU fallback
    m2 = jatsr_amd.JaT_AudioSR_V3(**recipe.CONFIGS["micro"], dropout=0.1, drop_path_rate=0.05)
    assert m2.blocks[0].drop_path_rate == 0.0 and m2.blocks[1].drop_path_rate == pytest.approx(0.05)   # linspace(0, rate, depth)
    assert all(b.dropout_rate == pytest.approx(0.1) for b in m2.blocks)


def _allreduce_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(8, dtype=torch.float32) * (rank + 1)
    w = T.allreduce_mean_(flat)
    out.put((rank, w, flat.tolist()))
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    """The step's only collective: SUM over ranks on the flat buffer; the 1/world factor is returned for the
    optimiser's unscale (jat_trainer_optim's loss_scale argument) instead of a second pass over the buffer."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, w, vals in got:
        assert w == 2 and vals == [3.0 * i for i in range(8)]
    assert T.allreduce_mean_(torch.ones(4)) == 1       # no process group: identity


def _exchange_worker(rank, world, port, out):
    import torch.distributed as dist
    from jatsr_amd.dist import exchange_sum_
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), JAT_EXCHANGE_MIN_BYTES="1024")   # small slices allowed here
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for n, mode in ((4096, "sync"), (4096 + 64, "async"), (1001, "fallback"), (64, "small")):   # 1001 % 2 != 0 / 256 B: all_reduce
        g = torch.Generator().manual_seed(100 + rank)
        buf = torch.randn(n, generator=g)
        ref = buf.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        works = exchange_sum_(buf, None, async_op=(mode == "async"))
        for w in works:
            w.wait()
        res[mode] = (bool(torch.equal(buf, ref)), float(buf.double().sum()))
    out.put((rank, res))
    dist.destroy_process_group()


def test_gradient_exchange_reduce_scatter_all_gather_world2_gloo():
    """The xGMI-shaped gradient exchange (jatsr_amd.dist.exchange_sum_): reduce-scatter of the rank's shard + all-gather,
    in place, equals the all-reduce bit for bit (two addends: order-free), on both ranks, sync and async forms; a length
    that does not divide by the world size takes the all_reduce fallback."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for mode in ("sync", "async", "fallback"):
        assert got[0][mode][0] and got[1][mode][0], mode
        assert got[0][mode][1] == got[1][mode][1], mode       # replicas hold identical sums


def _val_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 0 validated 3 batches, rank 1 validated 2 (a ragged DistributedSampler split): sums, not means, travel
    per_rank = {0: [1.0, 2.0, 3.0], 1: [10.0, 20.0]}[rank]
    acc = torch.zeros(8, dtype=torch.float64)
    for v in per_rank:
        acc[0] += v; acc[1] += 1; acc[2:7] += torch.tensor([v, 2 * v, 3 * v, 4 * v, 5 * v], dtype=torch.float64)
    avg, metrics = T.reduce_validation_sums(acc)
    out.put((rank, avg, metrics))
    dist.destroy_process_group()


def test_validation_reduce_world2_gloo():
    """The validation metrics of all ranks combine in ONE 8-float all-reduce of SUMS (train_ddp_v3mod2.py:1087-1096 issues
    seven 1-float ones): global average = sum of all batch losses / number of batches, identical on every rank."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_val_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (a, m)) for r, a, m in (q.get(timeout=120) for _ in procs))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    avg, metrics = got[0]
    assert avg == pytest.approx(36.0 / 5) and metrics["ms_loss"] == pytest.approx(3 * 36.0 / 5)
    assert metrics["total_latent_loss"] == pytest.approx(5 * 36.0 / 5)


def _loss_inputs(meta):
    B, C, Tn, salt = meta["B"], meta["C"], meta["T"], meta["salt"]
    pred = recipe.gaussian("loss_pred", (B, C, Tn), salt + 400)
    target = recipe.gaussian("loss_target", (B, C, Tn), salt + 401)
    lr = (0.7 * recipe.gaussian("loss_target", (B, C, Tn), salt + 401)
          + 0.5 * recipe.gaussian("loss_lr", (B, C, Tn), salt + 402)).astype(np.float32)
    return pred, target, lr


@pytest.mark.parametrize("name", ["train_loss_T24", "train_loss_T22", "train_loss_T9", "train_loss_T23", "train_loss_T1378"])
def test_latent_loss_oracle_matches_reference_classes(name):
    """MSE + latent perceptual loss of the v3mod2 trainer: the fixture ran the reference's own loss classes (taken from
    train_ddp_v3mod2.py with `ast`) under autograd in fp32; the numpy oracle (fp64) must agree to fp32-FFT accuracy."""
    from oracle import latent_loss_oracle as LO
    z, meta = load_golden(name)
    terms, dpred = LO.latent_loss(*_loss_inputs(meta), latent_weight=meta["lw"], freq_weight=meta["fw"],
                                  ms_weight=meta["mw"], consistency_weight=meta["cw"])
    for k in ("total", "mse", "freq", "ms", "consistency", "latent"):
        assert abs(terms[k] - float(z[k])) <= 2e-6 * abs(float(z[k])), k
    assert rel_l2(dpred, z["dpred"]) <= 1e-4


@pytest.mark.parametrize("name", ["train_micro_mod2_T24", "train_tiny_mod2_T128", "train_micro_mod2fw0_T24",
                                  "train_tiny_mod2fw0_T128"])
def test_train_oracle_v3mod2_step_matches_reference(name):
    """LayerNorm model + MSE + latent perceptual loss with the clean LR latent (train_ddp_v3mod2.py:854-896)."""
    z, meta = load_golden(name)
    cfg = recipe.CONFIGS[meta["cfg"]]
    C, B, Tn, salt = cfg["input_channels"], meta["B"], meta["T"], meta["salt"]
    sd = recipe.make_state_dict(cfg, "ln", salt)
    hr = recipe.gaussian("train_hr", (B, C, Tn), salt + 300).astype(np.float64)
    lr = recipe.gaussian("train_lr", (B, C, Tn), salt + 301).astype(np.float64)
    noise = recipe.gaussian("train_noise", (B, C, Tn), salt + 302).astype(np.float64)
    cn = 0.05 * recipe.gaussian("train_cnoise", (B, C, Tn), salt + 303).astype(np.float64)
    t = np.asarray(meta["t"], np.float32).astype(np.float64)
    tv = t.reshape(B, 1, 1)
    latent = dict(latent_weight=meta["lw"], freq_weight=meta["fw"], ms_weight=meta["mw"], consistency_weight=meta["cw"])
    loss, grads, pred = OT.TrainOracle(cfg, sd, "ln").loss_and_grads(tv * hr + (1 - tv) * noise, t, lr + cn, hr,
                                                                     latent=latent, cond_clean=lr)
    assert abs(loss - float(z["loss64"])) <= 1e-6 * float(z["loss64"])
    for k in meta["names"]:
        ref_l2 = float(z["gl2_" + k])
        assert abs(np.linalg.norm(grads[k]) - ref_l2) <= 2e-4 * max(ref_l2, 1e-30), k   # the reference runs its FFT terms in fp32
        assert rel_l2(gsub(grads[k], meta), z["g_" + k]) <= 5e-4, k
