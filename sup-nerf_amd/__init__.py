"""supnerf_amd -- MI355X (gfx950) implementation of SUP-NeRF's volumetric rendering hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's own function signatures
(``utils.render_rays_v2`` ..., ``renderer.NeRFRenderer``, ``SUPNeRF.forward``).  Device side: hand-written
HIP kernels in ``libsupnerf_hip.so`` behind the C ABI of ``include/supnerf_hip.h``.
"""
from . import _lib, ops, model, utils, renderer, synthetic, driver, io, trainer, scene  # noqa: F401
from . import binding  # noqa: F401
from .binding import install, uninstall, installed  # noqa: F401
from ._lib import SnrError  # noqa: F401
from .model import CodeNeRF, SUPNeRF  # noqa: F401
from .renderer import NeRFRenderer, render_rays_v3, volume_rendering3  # noqa: F401

__all__ = ["ops", "model", "utils", "renderer", "SnrError", "CodeNeRF", "SUPNeRF", "NeRFRenderer", "render_rays_v3",
           "volume_rendering3", "install", "uninstall", "installed"]
