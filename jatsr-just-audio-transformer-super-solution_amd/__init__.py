"""jatsr_amd — MI355X-native DiT flow-matching sampling path for JaTSR.

Host side keeps the reference's module API (`JaT_AudioSR_V3`, `DiTBlock_GQA`, `GroupedQueryAttention`,
`flow_matching_sample`, `crossfade_chunks`); all arithmetic on the path runs in hand-written gfx950 HIP
kernels behind the C ABI declared in `include/jat_hip.h` (`csrc/libjat_hip.so`).  There is no CPU fallback:
calling a compute entry point without the built library or without a GPU raises.
"""
from . import recipe  # noqa: F401  (numpy only)

__all__ = ["recipe"]


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import jatsr_amd.recipe` stays numpy-only
    if name in ("JaT_AudioSR_V3", "JaT_AudioSR_V2", "DiTBlock_GQA", "GroupedQueryAttention", "model"):
        from . import model as _m
        return _m if name == "model" else getattr(_m, name)
    if name in ("flow_matching_sample", "crossfade_chunks", "chunk_plan", "sample_long", "sampler"):
        from . import sampler as _s
        return _s if name == "sampler" else getattr(_s, name)
    if name in ("lib", "_lib"):
        from . import _lib
        return _lib
    raise AttributeError(name)
