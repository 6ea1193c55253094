"""ptmcmcsampler_amd -- MI355X-native engine for the Metropolis-Hastings inner loop of a
PTSampler-compatible parallel-tempering sampler.

Only the hot path lives here: HIP kernels + C ABI (``csrc/``, ``libptmi.so``), the ctypes
binding (``_lib``), the batched engine (``engine``), the temperature-block sharding
(``sharded``) and the ``PTSampler`` facade (``sampler``)."""
from .ladder import temperature_ladder  # noqa: F401

__all__ = ["temperature_ladder", "PTEngine", "PTSampler"]


def __getattr__(name):
    if name == "PTEngine":
        from .engine import PTEngine
        return PTEngine
    if name == "PTSampler":
        from .sampler import PTSampler
        return PTSampler
    raise AttributeError(name)
