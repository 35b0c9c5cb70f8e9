"""samplenerfro_amd — MI355X (gfx950) implementation of the SampleNeRFRO volumetric-rendering hot path.

The arithmetic lives in hand-written HIP kernels behind a C ABI (include/rnerf.h, samplenerfro_amd/csrc); this
package is the Python host that keeps the reference's call surface (rnerf/models.py, rnerf/utils.py, train.py).
"""
from . import _lib, prng, utils            # noqa: F401
from .utils import Rays, Stats, default_flags, render_image            # noqa: F401

__all__ = ["Rays", "Stats", "default_flags", "render_image", "prng", "utils"]


def __getattr__(name):
    # models/ops import torch lazily so that `import samplenerfro_amd` stays cheap for symbol checks
    if name in ("models", "ops", "synthetic", "grid", "train", "distributed"):
        import importlib
        return importlib.import_module(f".{name}", __name__)
    raise AttributeError(name)
