"""ctypes binding of libjat_hip.so (C ABI: include/jat_hip.h).

There is deliberately no fallback: if the library is not built, or a compute entry point is called
without a GPU, this raises — the product path never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The trainer queues its weight-gradient GEMMs on a second stream beside the backward's dX chain, and exchanges gradients on a
# third.  ROCm maps a process's streams onto four hardware queues by default: with a few more streams alive (samplers own one
# each) the second stream can land on the queue of the first, and the step then runs SLOWER than on one stream (measured: 59 ->
# 88 ms).  Eight queues keep them apart.  Only effective before the HIP runtime starts (importing torch does not start it); an
# explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# JAT_OPERAND_DTYPE=fp16 selects the fp16-operand build of the same kernels for the whole process (the v3mod2 trainer's
# autocast dtype, train_ddp_v3mod2.py:854); JAT_LIB_PATH overrides everything (A/B of two builds)
OPERAND_DTYPE = os.environ.get("JAT_OPERAND_DTYPE", "bf16").lower()
if OPERAND_DTYPE not in ("bf16", "fp16", "float16", "bfloat16"):
    raise ValueError(f"JAT_OPERAND_DTYPE must be bf16 or fp16, got {OPERAND_DTYPE!r}")
OPERAND_DTYPE = "fp16" if OPERAND_DTYPE in ("fp16", "float16") else "bf16"
LIB_PATH = os.environ.get("JAT_LIB_PATH") or os.path.join(
    _HERE, "csrc", "libjat_hip_fp16.so" if OPERAND_DTYPE == "fp16" else "libjat_hip.so")

JAT_OK, JAT_E_INVALID, JAT_E_HIP, JAT_E_STATE, JAT_E_SEQLEN = 0, -1, -2, -3, -4
NORM_RMS_W, NORM_LN_NOAFFINE = 0, 1


class JatConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "input_channels", "cond_channels", "patch_len", "hidden_size", "depth", "num_q_heads",
        "num_kv_heads", "bottleneck_dim", "mlp_hidden", "norm_mode")]


class JatTensorRef(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


# name -> (restype, argtypes); every symbol include/jat_hip.h declares
_VP, _I32, _I64, _F32, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "jat_last_error": (C.c_char_p, []),
    "jat_version": (C.c_int, []),
    "jat_operand_dtype": (C.c_int, []),
    "jat_model_create": (C.c_int, [C.POINTER(JatConfig), C.POINTER(_VP)]),
    "jat_model_destroy": (None, [_VP]),
    "jat_model_load_weights": (C.c_int, [_VP, C.POINTER(JatTensorRef), _I32, _VP]),
    "jat_model_set_switch": (C.c_int, [_VP, C.c_char_p, _I32]),
    "jat_model_workspace_bytes": (C.c_int, [_VP, _I32, _I32, C.POINTER(_SZ)]),
    "jat_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _VP, _SZ, _VP]),
    "jat_block_forward": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _I32, _I32, _VP, _SZ, _VP]),
    "jat_attn_forward": (C.c_int, [_VP, _I32, _VP, _VP, _I32, _I32, _VP, _SZ, _VP]),
    "jat_time_embed": (C.c_int, [_VP, _VP, _VP, _I32, _VP, _SZ, _VP]),
    "jat_sampler_create": (C.c_int, [_VP, _I32, _I32, _I32, _F32, C.POINTER(_VP)]),
    "jat_sampler_destroy": (None, [_VP]),
    "jat_sampler_run": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _VP]),
    "jat_sampler_info": (C.c_int, [_VP, _VP, _VP, _VP]),
    "jat_sampler_set_lengths": (C.c_int, [_VP, C.POINTER(_I32), _I32, _VP]),
    "jat_cfg_euler_step": (C.c_int, [_VP, _VP, _F32, _F32, _F32, _I32, _I32, _I32, _VP]),
    "jat_channel_affine": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    "jat_crossfade_pair": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _VP, _I32, _VP]),
    "jat_k_norm_modulate": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _VP, _I32, _I32, _I32, _I32, _VP]),
    "jat_k_gemm": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _I64, _I32, _I32, _VP]),
    "jat_k_gemm_fold": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _I64, _I32, _VP, _VP, _VP, _VP, _I32, _I32, _VP]),
    "jat_k_gemm_splitk": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "jat_k_gemm_wave_n": (C.c_int, [_I32]),
    "jat_k_qkv_attn": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _I32, _VP]),
    "jat_k_weight_grad": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _SZ, _VP]),
    "jat_k_attention": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "jat_k_recon_loss": (C.c_int, [_VP, _VP, _VP, _VP, _I64, C.c_double, _F32, _VP, _SZ, _VP]),
    "jat_k_cast_bf16": (C.c_int, [_VP, _VP, _I64, _VP]),
    "jat_k_latent_loss": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32] + [C.c_double] * 7 + [_F32, _VP, _SZ, _VP]),
    "jat_trainer_create": (C.c_int, [_VP, C.POINTER(JatTensorRef), _I32, _VP, _VP, _VP, _VP, _I64, _I32, _I32, _VP,
                                     C.POINTER(_VP)]),
    "jat_trainer_destroy": (None, [_VP]),
    "jat_trainer_workspace_bytes": (C.c_int, [_VP, C.POINTER(_SZ)]),
    "jat_trainer_repack": (C.c_int, [_VP, _VP]),
    "jat_trainer_prepare": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _F32, _I32, _VP, _VP, _VP, _VP]),
    "jat_trainer_set_grad_hook": (C.c_int, [_VP, _VP, _VP]),
    "jat_trainer_set_regularisers": (C.c_int, [_VP, C.POINTER(_F32), C.POINTER(_F32)]),
    "jat_trainer_set_latent_loss": (C.c_int, [_VP] + [C.c_double] * 7),
    "jat_trainer_set_charbonnier": (C.c_int, [_VP, C.c_double]),
    "jat_trainer_loss_terms": (C.c_int, [_VP, _VP, _VP]),
    "jat_trainer_fwd_bwd": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _F32, C.c_uint64, _VP, _VP, _VP]),
    "jat_trainer_optim": (C.c_int, [_VP, _F32, _F32, _F32, _F32, _F32, _F32, _F32, _I32, _VP, _VP]),
    "jat_prof_gemm_site": (C.c_int, [_VP, _I32, _I32]),
    "jat_prof_collect": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(_I32), C.POINTER(C.c_double), C.POINTER(_I32)]),
}

GRAD_HOOK = C.CFUNCTYPE(None, C.c_int64, C.c_int64, C.c_void_p)   # jat_trainer_set_grad_hook callback

_lib = None


class JatError(RuntimeError):
    pass


def lib():
    """Load libjat_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JatError(f"{LIB_PATH} not found: build it first (`python -c 'import __graft_entry__ as g; "
                           f"g.build()'` or `make -C {os.path.dirname(LIB_PATH)}`); there is no CPU fallback")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if not hasattr(h, name) and os.environ.get("JAT_LIB_ALLOW_MISSING"):
                continue      # A/B against an OLDER build through JAT_LIB_PATH (tools/sampler_ab.py): it lacks the newest entry points
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def operand_dtype() -> str:
    """'bf16' or 'fp16': what the loaded library rounds GEMM / attention operands to."""
    return "fp16" if lib().jat_operand_dtype() == 1 else "bf16"


def check(rc: int):
    """Translate a C-ABI return code into the exception the reference would raise at that point."""
    if rc == JAT_OK:
        return
    msg = lib().jat_last_error().decode("utf-8", "replace")
    if rc == JAT_E_SEQLEN:
        raise ValueError(msg)              # jat_audiosr_v3.py:451-452
    if rc == JAT_E_INVALID:
        raise ValueError(msg)
    raise JatError(f"libjat_hip error {rc}: {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise JatError("jatsr_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False and there is "
                       "no CPU fallback on the product path")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
