"""supnerf_amd -- MI355X (gfx950) implementation of SUP-NeRF's volumetric rendering hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's own function signatures
(``utils.render_rays_v2`` ..., ``renderer.NeRFRenderer``, ``SUPNeRF.forward``).  Device side: hand-written
HIP kernels in ``libsupnerf_hip.so`` behind the C ABI of ``include/supnerf_hip.h``.
"""
from . import _lib, ops  # noqa: F401
from ._lib import SnrError  # noqa: F401

__all__ = ["ops", "SnrError"]
