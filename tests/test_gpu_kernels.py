"""Per-kernel parity on the GPU, through the C ABI (include/jat_hip.h `jat_k_*`).

Each HIP kernel is compared with a plain PyTorch evaluation of the same op on the SAME bf16-rounded
operands (fp64 accumulate), so the only admissible differences are fp32 accumulation order and the final
bf16 rounding: tolerances are stated per test.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import jatsr_amd._lib as L  # noqa: E402

# operand dtype of the loaded library: bf16, or fp16 when the process runs with JAT_OPERAND_DTYPE=fp16 (the v3mod2 trainer's
# autocast dtype; tests/test_gpu_fp16.py re-runs this module that way).  fp16 has 3 more mantissa bits: same gates hold.
OP = torch.float16 if L.OPERAND_DTYPE == "fp16" else torch.bfloat16


def dev():
    L.require_gpu()
    return torch.device("cuda:0")


def bf16_bits(t):  # fp32 tensor -> (uint16-bits tensor as int16 view, rounded fp32 values)
    b = t.to(OP)
    return b, b.float()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def gen(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev())


def test_cast_bf16():
    x = gen((3, 1001), 1)
    out = torch.empty(x.shape, dtype=OP, device=x.device)
    L.check(L.lib().jat_k_cast_bf16(L.ptr(x), L.ptr(out), x.numel(), L.stream_ptr()))
    assert torch.equal(out, x.to(OP))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("D,M,ntok,shared", [(1280, 257, 64, False), (512, 64, 32, True), (256, 9, 3, False)])
def test_norm_modulate(mode, D, M, ntok, shared):
    B = (M + ntok - 1) // ntok
    x = gen((M, D), 2, 3.0) + 0.5
    w = 1 + 0.2 * gen((D,), 3)
    mod = gen((1 if shared else B, 2 * D), 4, 0.3)
    y = torch.empty(M, D, dtype=OP, device=x.device)
    shift, scale = mod[:, :D], mod[:, D:]
    L.check(L.lib().jat_k_norm_modulate(L.ptr(x), L.ptr(w), C.c_void_p(mod.data_ptr()),
                                        C.c_void_p(mod.data_ptr() + 4 * D), 0 if shared else 2 * D, L.ptr(y), M, D,
                                        ntok, mode, L.stream_ptr()))
    xd = x.double()
    if mode == 0:
        n = xd / torch.sqrt((xd * xd).mean(-1, keepdim=True) + 1e-6) * w.double()
    elif mode == 1:
        mu = xd.mean(-1, keepdim=True)
        n = (xd - mu) / torch.sqrt(((xd - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
    else:
        n = xd
    bidx = torch.zeros(M, dtype=torch.long, device=x.device) if shared else torch.arange(M, device=x.device) // ntok
    ref = n * (1 + scale.double()[bidx]) + shift.double()[bidx]
    # fp32 statistics, one bf16 rounding of the result (2^-9 relative)
    assert (y.double() - ref).abs().max() <= 2 ** -8 * ref.abs().max() + 1e-6
    assert rel(y, ref) < 3e-3


def test_norm_no_modulation():
    D, M = 1280, 37
    x = gen((M, D), 5)
    w = 1 + 0.2 * gen((D,), 6)
    y = torch.empty(M, D, dtype=OP, device=x.device)
    L.check(L.lib().jat_k_norm_modulate(L.ptr(x), L.ptr(w), None, None, 0, L.ptr(y), M, D, M, 0, L.stream_ptr()))
    xd = x.double()
    ref = xd / torch.sqrt((xd * xd).mean(-1, keepdim=True) + 1e-6) * w.double()
    assert rel(y, ref) < 3e-3


GEMM_SHAPES = [(128, 128, 64), (256, 512, 128), (1000, 1792, 1280), (56, 1536, 256), (1035, 1280, 5120),
               (384, 256, 8192),
               # N divisible by 160 / 320 / 448 (the wide tiles), K-tile counts 1, 2, 3 (pipeline prologue / tail paths)
               (300, 2240, 64), (300, 2240, 128), (500, 4480, 192),
               # more tiles than CUs: the one-block-per-CU variants walk several tiles per block (persistent launch)
               (4500, 4480, 128)]


# every live tile / pipeline variant of gemm.hip (ids are stable; the structures measured and retired in rounds 1-2 are
# rejected by launch_gemm: test_retired_gemm_variants_are_rejected); JAT_TEST_VARIANTS="31,32" narrows the sweep
LIVE_VARIANTS = [10, 18, 20, 21, 25, 26, 27, 28, 31, 32, 33, 34, 35, 36, 38, 39]
GEMM_VARIANTS = [int(v) for v in os.environ.get("JAT_TEST_VARIANTS", "").split(",") if v] or LIVE_VARIANTS


@pytest.mark.parametrize("variant", GEMM_VARIANTS)
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm(variant, M, N, K, epi):
    """C = A W^T (+bias) with epilogues fp32 / bf16 / bf16-GELU / gated residual; ragged M, asymmetric data."""
    A, Af = bf16_bits(gen((M, K), 10 + epi))
    W, Wf = bf16_bits(gen((N, K), 20 + epi, 1.0 / np.sqrt(K)))
    bias = gen((N,), 30, 0.1)
    ntok = 7 if M % 7 == 0 else (M if M < 64 else 23)
    B = (M + ntok - 1) // ntok
    gate = gen((B, N), 31, 0.5)
    ref = Af.double() @ Wf.double().T + bias.double()
    if epi == 0:
        out = torch.full((M, N), float("nan"), device=A.device)
    elif epi in (1, 2):
        out = torch.zeros((M, N), dtype=OP, device=A.device)
    else:
        out = gen((M, N), 32)
        x0 = out.clone()
    L.check(L.lib().jat_k_gemm(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out), M, N, K, epi, L.ptr(gate), N, ntok,
                               variant, L.stream_ptr()))
    torch.cuda.synchronize()
    if epi == 0:
        assert rel(out, ref) < 2e-6
        assert (out.double() - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))
    elif epi == 1:
        assert rel(out, ref) < 3e-3
        assert torch.equal(out, ref.float().to(OP)) or (out.float() - ref.float()).abs().max() <= \
            2 ** -8 * float(ref.abs().max())
    elif epi == 2:
        g = torch.nn.functional.gelu(ref)  # erf form, fp64
        assert (out.double() - g).abs().max() <= 2 ** -8 * float(g.abs().max()) + 1e-6
        assert rel(out, g) < 3e-3
    else:
        bidx = torch.arange(M, device=A.device) // ntok
        r = x0.double() + gate.double()[bidx] * ref
        assert rel(out, r) < 2e-6


def _fold_case(M, N, K, ntok, seed):
    A, Af = bf16_bits(gen((M, K), seed))
    W, Wf = bf16_bits(gen((N, K), seed + 1, 1.0 / np.sqrt(K)))
    bias = gen((N,), seed + 2, 0.1)
    B = (M + ntok - 1) // ntok
    gate = gen((B, N), seed + 3, 0.5)
    x0 = gen((M, N), seed + 4, 2.0)
    hi0 = x0.to(OP)
    lo0 = (x0 - hi0.float()).to(OP)
    return A, Af, W, Wf, bias, gate, hi0, lo0


@pytest.mark.parametrize("variant", [20, 25, 32])
@pytest.mark.parametrize("M,N,K,ntok", [(300, 1280, 256, 128), (512, 1280, 128, 128), (300, 1280, 192, 23), (700, 1280, 64, 345)])
@pytest.mark.parametrize("epi", [0, 3])
def test_gemm_fold_producer(variant, M, N, K, ntok, epi):
    """Split-residual epilogue of the sampler's folded norms (jat_k_gemm_fold): x_new = acc + bias (epilogue 0) or
    (hi + lo) + gate (acc + bias) (epilogue 3, jat_audiosr_v3.py:300,306) kept as two 16-bit planes, plus the row partial
    sums of x_new^2 the consumer's rstd is built from; ragged M, wave tiles inside one sample / across two / across many."""
    A, Af, W, Wf, bias, gate, hi0, lo0 = _fold_case(M, N, K, ntok, 200 + epi)
    if N % (L.lib().jat_k_gemm_wave_n(variant) or 1) != 0:
        pytest.skip("tile does not divide N")
    wn = L.lib().jat_k_gemm_wave_n(variant)
    hi, lo = hi0.clone(), lo0.clone()
    part = torch.full((M, N // wn), float("nan"), device=A.device)
    L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), None, M, N, K, epi, L.ptr(gate), N, ntok, L.ptr(hi),
                                    L.ptr(lo), L.ptr(part), None, 0, variant, L.stream_ptr()))
    torch.cuda.synchronize()
    y = Af.double() @ Wf.double().T + bias.double()
    if epi == 3:
        bidx = torch.arange(M, device=A.device) // ntok
        ref = (hi0.double() + lo0.double()) + gate.double()[bidx] * y
    else:
        ref = y
    got = hi.double() + lo.double()
    # hi = round(x), lo = round(x - hi): 16 significant bits (bf16) / 22 (fp16)
    assert (got - ref).abs().max() <= 2 ** -15 * float(ref.abs().max()) + 1e-6
    # hi alone is x rounded to the operand dtype: it is the next GEMM's A operand as it stands
    assert ((hi.double() - ref).abs() <= 2 ** -8 * ref.abs() + 1e-6).all()
    assert rel(part.double().sum(1), (ref * ref).sum(1)) < 1e-5


@pytest.mark.parametrize("M,N,K,ntok", [(224, 160, 64, 128), (448, 1280, 128, 128), (448, 1280, 192, 23), (672, 320, 256, 345),
                                        (7168, 1280, 1280, 128), (7168, 1280, 5120, 128)])
@pytest.mark.parametrize("epi", [0, 3])
def test_gemm_kpair_kernel(M, N, K, ntok, epi):
    """Variant 39 (gemm_kpair_kernel: 224 x 160 tile, two wave groups on the two k-steps of every K-tile, three LDS stages,
    partial accumulators exchanged through LDS) on the split-residual producer epilogues: against fp64, 1 / 2 / 3 / 4 / 20 / 80
    K-tiles (every prologue and tail path), wave tiles inside one sample and across several; and deterministic."""
    A, Af, W, Wf, bias, gate, hi0, lo0 = _fold_case(M, N, K, ntok, 240 + epi)
    res = []
    for _ in range(2):
        hi, lo = hi0.clone(), lo0.clone()
        part = torch.full((M, N // 80), float("nan"), device=A.device)
        L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), None, M, N, K, epi, L.ptr(gate), N, ntok, L.ptr(hi),
                                        L.ptr(lo), L.ptr(part), None, 0, 39, L.stream_ptr()))
        torch.cuda.synchronize()
        res.append((hi, lo, part))
    assert all(torch.equal(a, b) for a, b in zip(*res))
    hi, lo, part = res[0]
    y = Af.double() @ Wf.double().T + bias.double()
    if epi == 3:
        bidx = torch.arange(M, device=A.device) // ntok
        ref = (hi0.double() + lo0.double()) + gate.double()[bidx] * y
    else:
        ref = y
    assert ((hi.double() + lo.double()) - ref).abs().max() <= 2 ** -15 * float(ref.abs().max()) + 1e-5
    assert ((hi.double() - ref).abs() <= 2 ** -8 * ref.abs() + 1e-5).all()
    assert rel(part.double().sum(1), (ref * ref).sum(1)) < 1e-5


@pytest.mark.parametrize("variant,M,N,K,ksplit", [(39, 224, 160, 384, 2), (39, 448, 320, 1280, 2), (39, 3584, 1280, 5120, 2),
                                                  (39, 224, 1280, 2560, 4), (20, 200, 256, 512, 2), (20, 3584, 1280, 5120, 2)])
def test_gemm_splitk_slices(variant, M, N, K, ksplit):
    """Split-K slices (jat_k_gemm_splitk): slice z = the product over its K columns alone, for the plain 128 x 128 tile and for
    the 224 x 160 k-step-pair tile the half-size forward (BASELINE configs[1], M = 3584) cuts fc2 with; every slice against
    fp64, the rest of the workspace untouched, deterministic."""
    A, Af = bf16_bits(gen((M, K), 260))
    W, Wf = bf16_bits(gen((N, K), 261, 1.0 / np.sqrt(K)))
    res = []
    for _ in range(2):
        parts = torch.full((ksplit + 1, M, N), float("nan"), device=A.device)
        L.check(L.lib().jat_k_gemm_splitk(L.ptr(A), L.ptr(W), L.ptr(parts), M, N, K, ksplit, variant, L.stream_ptr()))
        torch.cuda.synchronize()
        res.append(parts)
    assert torch.equal(res[0][:ksplit], res[1][:ksplit]) and torch.isnan(res[0][ksplit]).all()
    ks = K // ksplit
    for z in range(ksplit):
        ref = Af[:, z * ks:(z + 1) * ks].double() @ Wf[:, z * ks:(z + 1) * ks].double().T
        assert (res[0][z].double() - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("variant", [31, 36])
@pytest.mark.parametrize("np_", [4, 8, 16])
@pytest.mark.parametrize("epi", [1, 2])
def test_gemm_fold_consumer(variant, np_, epi):
    """Consumer side: accumulator rows scaled by rsqrt(sum_j part[m][j] / K + 1e-6) before the bias (the RMSNorm statistic
    of jat_audiosr_v3.py:297,303 applied after the matmul), bf16 / GELU outputs; 36 = 31 + pipelined epilogue, bit-identical."""
    M, N, K, ntok = 500, 2240, 192, 128
    A, Af = bf16_bits(gen((M, K), 220))
    W, Wf = bf16_bits(gen((N, K), 221, 1.0 / np.sqrt(K)))
    bias = gen((N,), 222, 0.1)
    part = gen((M, np_), 223).abs() * K / np_ + 0.1
    out = torch.zeros((M, N), dtype=OP, device=A.device)
    L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out), M, N, K, epi, None, 0, ntok, None, None, None,
                                    L.ptr(part), np_, variant, L.stream_ptr()))
    torch.cuda.synchronize()
    rstd = torch.rsqrt(part.double().sum(1, keepdim=True) / K + 1e-6)
    ref = rstd * (Af.double() @ Wf.double().T) + bias.double()
    if epi == 2:
        ref = torch.nn.functional.gelu(ref)
    assert rel(out, ref) < 3e-3
    if variant == 36:
        out31 = torch.zeros_like(out)
        L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out31), M, N, K, epi, None, 0, ntok, None, None,
                                        None, L.ptr(part), np_, 31, L.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(out, out31)


@pytest.mark.parametrize("M,N,K", [(448, 640, 128), (3808, 5120, 64), (3808, 5120, 128), (3808, 5120, 192), (7168, 5120, 1280),
                                   (57568, 320, 256)])
@pytest.mark.parametrize("epi,use_part,use_bias", [(2, True, True), (1, True, False), (2, False, True), (1, False, False)])
def test_gemm_persistent_two_tile_kernel(M, N, K, epi, use_part, use_bias):
    """Variant 38 (gemm_persist_kernel: one block per CU walks two 224 x 320 tiles, the second tile's first K-tile, bias slice
    and row statistics prefetched under the first tile's epilogue) is bit-identical to the one-tile kernel (variant 36) and
    right against fp64: one tile per block, 16 / all blocks with two tiles, 1 / 2 / 3 / 20 K-tiles (every prologue / tail path),
    with and without the consumer's row statistics and bias."""
    A, Af = bf16_bits(gen((M, K), 230))
    W, Wf = bf16_bits(gen((N, K), 231, 1.0 / np.sqrt(K)))
    bias = gen((N,), 232, 0.1) if use_bias else None
    part = (gen((M, 16), 233).abs() * K / 16 + 0.1) if use_part else None
    outs = []
    for variant in (36, 38):
        out = torch.full((M, N), float("nan"), device=A.device).to(OP)
        L.check(L.lib().jat_k_gemm_fold(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(out), M, N, K, epi, None, 0, 128, None, None, None,
                                        L.ptr(part), 16 if use_part else 0, variant, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    if M * N <= 3808 * 5120:
        ref = Af.double() @ Wf.double().T
        if use_part:
            ref = ref * torch.rsqrt(part.double().sum(1, keepdim=True) / K + 1e-6)
        if use_bias:
            ref = ref + bias.double()
        if epi == 2:
            ref = torch.nn.functional.gelu(ref)
        assert rel(outs[1], ref) < 3e-3


@pytest.mark.parametrize("tokens,out,inn,ksplit", [
    (64, 128, 128, 1), (1, 128, 128, 1), (100, 128, 256, 1), (1000, 384, 256, 1), (1000, 384, 256, 4),
    (2583, 256, 640, 3), (9660, 1280, 1280, 4), (9660, 5120, 1280, 1), (6182, 1280, 5120, 2),
    # 256 x 256 tiles (out * in >= 2^20): automatic split, ragged and < 64-token batches, one K-tile per slice
    (9660, 1280, 1280, 0), (9660, 3840, 1280, 0), (2583, 1024, 1024, 0), (37, 1024, 1024, 1), (4096, 1024, 1280, 7),
    (130, 1024, 1024, 3), (9660, 5120, 1280, 0)])
def test_weight_grad(tokens, out, inn, ksplit):
    """dW = dY^T X and db = column sums of dY straight from the token-major operands (gemm_tn.hip): ragged token counts
    (last K-tile zero-filled in LDS), split-K summed in order, no transposed copies.  The buffers are allocated exactly
    `tokens` rows long inside a poisoned arena: whatever follows them must not leak into the sums."""
    arena_y = torch.full(((tokens + 64) * out,), float("nan"), device=dev()).to(OP)
    arena_x = torch.full(((tokens + 64) * inn,), float("nan"), device=dev()).to(OP)
    dY = arena_y[:tokens * out].view(tokens, out)
    X = arena_x[:tokens * inn].view(tokens, inn)
    dY.copy_(gen((tokens, out), 40, 0.05).to(OP))
    X.copy_((gen((tokens, inn), 41) + 0.25).to(OP))
    dW = torch.full((out, inn), float("nan"), device=dev())
    db = torch.full((out,), float("nan"), device=dev())
    work = torch.full((64 + 16 * out * inn + 32 * out,), float("nan"), device=dev())
    L.check(L.lib().jat_k_weight_grad(L.ptr(dY), L.ptr(X), L.ptr(dW), L.ptr(db), tokens, out, inn, ksplit, L.ptr(work),
                                      work.numel() * 4, L.stream_ptr()))
    torch.cuda.synchronize()
    ref = dY.double().T @ X.double()
    assert rel(dW, ref) < 3e-6
    assert (dW.double() - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))
    rb = dY.double().sum(0)
    assert (db.double() - rb).abs().max() < 2e-5 * max(1.0, float(rb.abs().max()))
    # bit-reproducible: fixed-order partial sums, no atomics
    dW2 = torch.empty_like(dW)
    L.check(L.lib().jat_k_weight_grad(L.ptr(dY), L.ptr(X), L.ptr(dW2), None, tokens, out, inn, ksplit, L.ptr(work),
                                      work.numel() * 4, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dW, dW2)


def test_weight_grad_rejects_unsupported_shapes():
    z = torch.zeros((64, 192), device=dev()).to(OP)
    o = torch.zeros((192, 192), device=dev())
    assert L.lib().jat_k_weight_grad(L.ptr(z), L.ptr(z), L.ptr(o), None, 64, 192, 192, 1, None, 0, L.stream_ptr()) != 0


def test_retired_gemm_variants_are_rejected():
    A = torch.zeros(128, 64, dtype=OP, device=dev())
    out = torch.zeros(128, 128, device=dev())
    for v in [v for v in range(-1, 40) if v not in LIVE_VARIANTS]:
        rc = L.lib().jat_k_gemm(L.ptr(A), L.ptr(A), None, L.ptr(out), 128, 128, 64, 0, None, 0, 128, v, L.stream_ptr())
        assert rc != 0, v


def _attention_ref(q, k, v, B, N, Hq, Hkv):
    g = Hq // Hkv
    Q = q.double().view(B, N, Hq, 64).transpose(1, 2)
    K = k.double().view(B, N, Hkv, 64).transpose(1, 2).repeat_interleave(g, dim=1)
    V = v.double().view(B, N, Hkv, 64).transpose(1, 2).repeat_interleave(g, dim=1)
    S = Q @ K.transpose(-1, -2) / 8.0
    O = torch.softmax(S, -1) @ V
    return O.transpose(1, 2).reshape(B * N, Hq * 64)


@pytest.mark.parametrize("B,N,Hq,Hkv", [(2, 128, 20, 4), (1, 345, 20, 4), (3, 6, 4, 2), (1, 1024, 8, 4), (2, 70, 4, 4)])
def test_attention(B, N, Hq, Hkv):
    """softmax(q k^T/8) v with GQA head sharing; N ragged w.r.t. the 64-key block and the 128-query block."""
    npad = (N + 63) // 64 * 64
    q, qf = bf16_bits(gen((B * N, Hq * 64), 40, 1.5))
    k, kf = bf16_bits(gen((B * N, Hkv * 64), 41, 1.5))
    v, vf = bf16_bits(gen((B * N, Hkv * 64), 42))
    vt = torch.zeros(B, Hkv, 64, npad, dtype=OP, device=q.device)
    vt[:, :, :, :N] = v.view(B, N, Hkv, 64).permute(0, 2, 3, 1)
    o = torch.zeros(B * N, Hq * 64, dtype=OP, device=q.device)
    L.check(L.lib().jat_k_attention(L.ptr(q), L.ptr(k), L.ptr(vt), L.ptr(o), B, N, Hq, Hkv, npad, L.stream_ptr()))
    ref = _attention_ref(qf, kf, vf, B, N, Hq, Hkv)
    # P is rounded to bf16 before PV and the output once more: ~2^-8 relative to the value scale
    assert rel(o, ref) < 6e-3
    assert (o.double() - ref).abs().max() < 2e-2 * float(ref.abs().max())


def test_attention_spiky_rows():
    """Online-softmax rescale across key blocks: one key in the LAST block dominates a query row."""
    B, N, Hq, Hkv = 1, 256, 4, 2
    npad = 256
    qx = gen((B * N, Hq * 64), 50)
    kx = gen((B * N, Hkv * 64), 51)
    kx[200, :64] = qx[5, :64] * 4.0  # query 5 of head 0 / kv-head 0 spikes at key 200 (4th block)
    kx[3, 64:] = qx[77, 128:192] * 4.0  # head 2 -> kv head 1 spikes in the first block
    q, qf = bf16_bits(qx)
    k, kf = bf16_bits(kx)
    v, vf = bf16_bits(gen((B * N, Hkv * 64), 52))
    vt = v.view(B, N, Hkv, 64).permute(0, 2, 3, 1).contiguous()
    o = torch.zeros(B * N, Hq * 64, dtype=OP, device=q.device)
    L.check(L.lib().jat_k_attention(L.ptr(q), L.ptr(k), L.ptr(vt), L.ptr(o), B, N, Hq, Hkv, npad, L.stream_ptr()))
    ref = _attention_ref(qf, kf, vf, B, N, Hq, Hkv)
    assert rel(o, ref) < 6e-3
    assert (o.double() - ref).abs().max() < 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("use_cfg,t", [(True, 0.3), (False, 0.5), (True, 0.9995)])
def test_cfg_euler_step(use_cfg, t):
    """Bit-exact against the reference expression order (infer_test_v3m2.py:161-179) in fp32."""
    B, Cc, T = 2, 8, 37
    scale = 3.0 if use_cfg else 1.0
    xp = gen((2 * B if use_cfg else B, Cc, T), 60)
    z = gen((B, Cc, T), 61)
    z0 = z.clone()
    dt = 0.02
    L.check(L.lib().jat_cfg_euler_step(L.ptr(xp), L.ptr(z), scale, t, dt, B, Cc, T, L.stream_ptr()))
    x = xp[B:] + scale * (xp[:B] - xp[B:]) if use_cfg else xp
    tt = torch.tensor(t, dtype=torch.float32)
    if t < 0.999:
        ref = z0 + (x - z0) / (1 - tt + 1e-5).to(x.device) * torch.tensor(dt, dtype=torch.float32, device=x.device)
    else:
        ref = x
    assert torch.allclose(z, ref, rtol=0, atol=2e-6)
    assert rel(z, ref) < 1e-6


def test_channel_affine_roundtrip():
    B, Cc, T = 2, 32, 50
    x = gen((B, Cc, T), 70)
    mean, std = gen((Cc,), 71), gen((Cc,), 72).abs() + 0.5
    y = torch.empty_like(x)
    L.check(L.lib().jat_channel_affine(L.ptr(x), L.ptr(mean), L.ptr(std), L.ptr(y), B, Cc, T, 0, L.stream_ptr()))
    assert torch.allclose(y, (x - mean.view(1, -1, 1)) / std.view(1, -1, 1), atol=1e-6)
    x2 = torch.empty_like(x)
    L.check(L.lib().jat_channel_affine(L.ptr(y), L.ptr(mean), L.ptr(std), L.ptr(x2), B, Cc, T, 1, L.stream_ptr()))
    assert torch.allclose(x2, x, atol=1e-5)
