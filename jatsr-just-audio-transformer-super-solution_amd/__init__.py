"""jatsr_amd — MI355X-native DiT flow-matching sampling path for JaTSR.

Host side keeps the reference's module API (`JaT_AudioSR_V3`, `DiTBlock_GQA`, `GroupedQueryAttention`,
`flow_matching_sample`, `crossfade_chunks`); all arithmetic on the path runs in hand-written gfx950 HIP
kernels behind the C ABI declared in `include/jat_hip.h` (`csrc/libjat_hip.so`).  There is no CPU fallback:
calling a compute entry point without the built library or without a GPU raises.
"""
import importlib as _importlib

from . import recipe  # noqa: F401  (numpy only)

_LAZY = {
    "JaT_AudioSR_V3": "model", "JaT_AudioSR_V2": "model", "DiTBlock_GQA": "model",
    "GroupedQueryAttention": "model", "load_model": "model",
    "flow_matching_sample": "sampler", "crossfade_chunks": "sampler", "chunk_plan": "sampler", "chunk_groups": "sampler",
    "sample_long": "sampler", "Sampler": "sampler", "channel_affine": "sampler",
    "load_latent_file": "io", "save_latent_file": "io", "load_stats": "io",
    "Trainer": "train", "u_shaped_timestep_sampling": "train", "get_lr": "train", "GradScaler": "train",
}
__all__ = ["recipe"] + sorted(_LAZY)


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import jatsr_amd.recipe` stays numpy-only
    if name in _LAZY:
        return getattr(_importlib.import_module(f"{__name__}.{_LAZY[name]}"), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
