"""MI355X-native hot path of correlated volumetric photon mapping:
Woodcock photon trace -> radix sort / uniform-grid bin -> per-cell density gather,
hand-written HIP for gfx950 behind a C-ABI (include/cpm/cpm.h).

The directory name carries hyphens (it is the reference's name + ``_amd``), so import it
through the ``cpm_amd`` shim at the repository root or ``importlib.import_module``.
Importing this package does not load the shared library; ``binding.load_library()`` /
``binding.Context()`` do, and fail loudly when it has not been built or no GPU is present.
"""
from . import binding, synthetic  # noqa: F401

__all__ = ["binding", "synthetic", "pipeline", "sharding", "build", "hostlayer"]


def __getattr__(name):
    if name in ("pipeline", "build", "sharding", "hostlayer"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
