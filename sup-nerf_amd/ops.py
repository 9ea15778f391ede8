"""Tensor-level operators over the C ABI: torch allocates, the HIP library computes.

Everything here requires fp32 tensors on an AMD GPU (``tensor.is_cuda`` under PyTorch-ROCm).
There is no CPU/eager fallback -- a CPU tensor raises."""
import copy
import ctypes as C
import math
import threading
import warnings
import weakref
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import BF16X3, BOX_DETACH, FP32, METRIC_Z, WHITE_BKGD, Z_BOX, Z_PER_OBJECT, Z_PER_RAY, Z_SHARED, RenderArgs, SnrError, check

_AUTO_DOWNGRADES = set()
IDENTITY_FRAME = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)
PRECISIONS = {"fp32": FP32, "bf16x3": BF16X3}


def resolve_precision(precision, shape_blocks, texture_blocks, points_per_obj, backward=False) -> int:
    """'fp32' (exact fp32 MFMA), 'bf16x3' (the split kernels: every fp32 operand as two 16-bit pieces, three MFMAs per product, several times
    faster -- fp16 pieces in the forward chain, ~2^-22 per product, bf16 pieces in the backward chain, ~2^-17) or 'auto'
    (bf16x3 where the kernel supports the configuration, else fp32).  Asking for 'bf16x3' where it is unsupported raises.
    A pair (forward, backward) names the two launches apart, e.g. ("fp32", "bf16x3"): the reference's forward values bit for bit in
    its own arithmetic, the gradient on the split-bf16 kernel (it applies the ReLU pattern the forward saved; 2^-17 per product) --
    ``backward`` picks the member.  (A training triple (forward chain, backward chain, products) resolves like its first two.)"""
    if isinstance(precision, tuple):
        if len(precision) not in (2, 3):
            raise SnrError(f"a precision tuple is (forward, backward) or (forward chain, backward chain, products), got {precision!r}")
        precision = precision[1 if backward else 0]
    if isinstance(precision, int) and not isinstance(precision, bool):
        return precision
    if precision is None or precision == "auto":
        ok = _lib.lib().snr_precision_supported(BF16X3, shape_blocks, texture_blocks, int(points_per_obj))
        if not ok:
            key = (shape_blocks, texture_blocks, int(points_per_obj) % 32 == 0)
            if key not in _AUTO_DOWNGRADES:          # say it once per configuration: 'auto' is several times slower here
                _AUTO_DOWNGRADES.add(key)
                warnings.warn(f"supnerf_amd: precision 'auto' runs the exact fp32 kernels for shape_blocks={shape_blocks}, texture_blocks={texture_blocks}, "
                              f"{points_per_obj} points per object: the split kernels need shape_blocks + texture_blocks <= 4 and whole 32-point tiles "
                              "per object", RuntimeWarning, stacklevel=3)
        return BF16X3 if ok else FP32
    if precision not in PRECISIONS:
        raise SnrError(f"unknown precision {precision!r}")
    code = PRECISIONS[precision]
    if not _lib.lib().snr_precision_supported(code, shape_blocks, texture_blocks, int(points_per_obj)):
        raise SnrError(f"precision {precision!r} is not available for shape_blocks={shape_blocks}, texture_blocks={texture_blocks}, "
                       f"{points_per_obj} points per object (needs <= 4 blocks in total and whole 32-point tiles per object)")
    return code


PRECISION_NAMES = {FP32: "fp32", BF16X3: "bf16x3"}
FP16_MAX = 65504.0
# what the range guard of precision "auto" accepts between the split forward and the exact-fp32 forward of the same launch: |a - b| <=
# RANGE_TOL * max(1, |b|) per value (measured distance of a healthy decoder: 6e-8 .. 1.6e-6)
RANGE_TOL = 1e-5


def precision_pair(precision, shape_blocks, texture_blocks, points_per_obj):
    """(forward, backward) arithmetic names a launch pair with this request would run in."""
    f = resolve_precision(precision, shape_blocks, texture_blocks, points_per_obj)
    b = resolve_precision(precision, shape_blocks, texture_blocks, points_per_obj, backward=True)
    return PRECISION_NAMES[f], PRECISION_NAMES[b]


def split_supported(shape_blocks, texture_blocks, points_per_obj) -> bool:
    return bool(_lib.lib().snr_precision_supported(BF16X3, shape_blocks, texture_blocks, int(points_per_obj)))


def outputs_disagree(split_outs, exact_outs, tol=RANGE_TOL):
    """0-dim device tensor: how many values of the split-arithmetic outputs are non-finite where the exact ones are finite, or further
    than ``tol * max(1, |exact|)`` from them (one host read decides; the range guard of ``auto``)."""
    bad = None
    for a, b in zip(split_outs, exact_outs):
        a, b = a.detach().reshape(-1), b.detach().reshape(-1)
        fin = torch.isfinite(b)
        n = ((~torch.isfinite(a)) & fin).sum() + (((a - b).abs() > tol * b.abs().clamp_min(1.0)) & fin).sum()
        bad = n if bad is None else bad + n
    return bad


def clamped_weight_count(params) -> torch.Tensor:
    """0-dim device tensor: entries of the per-point weights beyond the fp16 range -- the split forward packs them clamped to +-65504."""
    return sum((p.detach().abs() > FP16_MAX).sum() for p in params)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise SnrError("supnerf_amd operators need tensors on the GPU (no CPU fallback)")


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _p(t: Optional[torch.Tensor]):
    """Device address for the C ABI.  The ABI takes dense buffers and cannot know strides (include/supnerf_hip.h: "contiguous"), so
    EVERY pointer that crosses it goes through here and a tensor that is not a dense GPU buffer raises instead of being read out
    of bounds (round 2's fault: ``get_rays``' origins are a stride-0 view of the pose's three numbers)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise SnrError("supnerf_amd operators need tensors on the GPU (no CPU fallback)")
    if not t.is_contiguous():
        raise SnrError(f"internal: a non-contiguous tensor (shape {tuple(t.shape)}, strides {t.stride()}) reached the C ABI; "
                       "operands must be made dense with _f32c first")
    return C.c_void_p(t.data_ptr())


def _p_rows(t: torch.Tensor):
    """Address of a 2-D row-major operand that travels WITH its leading dimension (snr_weight_grad's G, X, dW: a column block of a wider
    buffer is fine there); only the rows must be dense."""
    if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32:
        raise SnrError(f"expected a 2-D fp32 row-major GPU tensor, got shape {tuple(t.shape)} strides {t.stride()} {t.dtype} on {t.device}")
    return C.c_void_p(t.data_ptr())


def _ptr(t: Optional[torch.Tensor], dtype=torch.float32) -> int:
    """Like ``_p`` for struct fields (plain integer address), with the element type checked too."""
    if t is None:
        return 0
    if t.dtype != dtype:
        raise SnrError(f"internal: a {t.dtype} tensor reached the C ABI where {dtype} is expected")
    return _p(t).value or 0


def raw_stream(dev) -> int:
    """The current HIP stream of `dev` as an integer handle.  torch.cuda.current_stream() builds a Python Stream object per call (5 us; the
    public render functions ask ~18 times per optimiser iteration); the raw getter behind it does not."""
    d = dev if isinstance(dev, torch.device) else torch.device(dev)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    try:
        return int(torch._C._cuda_getCurrentRawStream(idx))
    except AttributeError:      # (a torch build without the private getter)
        return int(torch.cuda.current_stream(d).cuda_stream)


def _stream(dev):
    return C.c_void_p(raw_stream(dev))


# ------------------------------------------------------------------------------------ weights
def per_point_tensor_names(shape_blocks: int, texture_blocks: int) -> Sequence[str]:
    """State-dict keys (reference naming, src/model_supnerf.py:184-199) of the per-point layers in the
    order snr_pack_weights expects them."""
    names = ["encoding_xyz.0"] + [f"shape_layer_{j}.0" for j in range(1, shape_blocks + 1)]
    names += ["encoding_shape", "sigma.0", "encoding_viewdir.0"]
    names += [f"texture_layer_{j}.0" for j in range(1, texture_blocks + 1)] + ["rgb.0", "rgb.2"]
    out = []
    for n in names:
        out += [n + ".weight", n + ".bias"]
    return out


def check_decoder_shapes(params: Dict[str, torch.Tensor], shape_blocks: int, texture_blocks: int):
    """The kernels are built for the decoder of every shipped config: W = latent = 256, L_xyz = 10, L_dir = 4."""
    want = {"encoding_xyz.0.weight": (256, 63), "encoding_shape.weight": (256, 256), "sigma.0.weight": (1, 256),
            "encoding_viewdir.0.weight": (256, 283), "rgb.0.weight": (128, 256), "rgb.2.weight": (3, 128)}
    for j in range(1, shape_blocks + 1):
        want[f"shape_layer_{j}.0.weight"] = (256, 256)
    for j in range(1, texture_blocks + 1):
        want[f"texture_layer_{j}.0.weight"] = (256, 256)
    for k, shp in want.items():
        if k not in params or tuple(params[k].shape) != shp:
            got = tuple(params[k].shape) if k in params else None
            raise SnrError(f"unsupported decoder: {k} is {got}, the gfx950 kernels need {shp} "
                           "(W=256, latent_dim=256, num_xyz_freq=10, num_dir_freq=4)")


def pack_weights(params: Dict[str, torch.Tensor], shape_blocks: int, texture_blocks: int) -> torch.Tensor:
    """nn.Linear tensors (reference state-dict names) -> the kernels' packed stream buffer."""
    check_decoder_shapes(params, shape_blocks, texture_blocks)
    names = per_point_tensor_names(shape_blocks, texture_blocks)
    ts = [_f32c(params[n].detach()) for n in names]
    _need_gpu(*ts)
    dev = ts[0].device
    nbytes = _lib.lib().snr_packed_bytes(shape_blocks, texture_blocks)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    arr = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    with torch.cuda.device(dev):
        check(_lib.lib().snr_pack_weights(arr, len(ts), shape_blocks, texture_blocks, _p(packed), _stream(dev)), "snr_pack_weights")
    return packed


# ------------------------------------------------------------------------------------ composite
def composite_fwd(sigmas, rgbs, z_vals, z_mode, white_bkgd, rays_per_obj=0):
    sigmas, rgbs, z_vals = _f32c(sigmas), _f32c(rgbs), _f32c(z_vals)
    _need_gpu(sigmas, rgbs, z_vals)
    S = rgbs.shape[-2]
    n_rays = rgbs.numel() // (3 * S) if rgbs.numel() else 0
    dev = rgbs.device
    rgb = torch.empty(n_rays, 3, device=dev)
    depth = torch.empty(n_rays, device=dev)
    acc = torch.empty(n_rays, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_composite_fwd(_p(sigmas), _p(rgbs), _p(z_vals), z_mode, WHITE_BKGD if white_bkgd else 0, n_rays,
                                           rays_per_obj, S, _p(rgb), _p(depth), _p(acc), _stream(dev)), "snr_composite_fwd")
    return rgb, depth, acc


def scene_composite(sigmas, rgbs, z_vals, white_bkgd=True, run_length=0):
    """Per-pixel depth merge of n = Nb*S samples + composite (scripts/demo.py:555-565): sigmas, z_vals (P, n); rgbs (P, n, 3)
    -> rgb (P,3), depth (P), acc_trans (P).  Inference only (the reference runs it under no_grad).  ``run_length`` = S tells the kernel that
    every object's S samples are contiguous and ascending, so it merges the lists instead of rank-sorting (verified per pixel, see the header)."""
    sigmas, rgbs, z_vals = _f32c(sigmas), _f32c(rgbs), _f32c(z_vals)
    _need_gpu(sigmas, rgbs, z_vals)
    if sigmas.dim() != 2 or rgbs.shape != (*sigmas.shape, 3) or z_vals.shape != sigmas.shape:
        raise SnrError(f"scene_composite: expected sigmas/z (P,n) and rgbs (P,n,3), got {tuple(sigmas.shape)}, {tuple(z_vals.shape)}, {tuple(rgbs.shape)}")
    P, n = sigmas.shape
    dev = sigmas.device
    rgb = torch.empty(P, 3, device=dev); depth = torch.empty(P, device=dev); acc = torch.empty(P, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_scene_composite_fwd(_p(sigmas), _p(rgbs), _p(z_vals), P, n, int(run_length), WHITE_BKGD if white_bkgd else 0,
                                                 _p(rgb), _p(depth), _p(acc), _stream(dev)), "snr_scene_composite_fwd")
    return rgb, depth, acc


def composite_bwd(sigmas, rgbs, z_vals, z_mode, white_bkgd, rays_per_obj, d_rgb, d_depth, d_acc, need_dz):
    sigmas, rgbs, z_vals = _f32c(sigmas), _f32c(rgbs), _f32c(z_vals)
    # dense copies are NAMED so that they live until the launch is enqueued: `_p(_f32c(t))` inside the argument list frees the temporary before
    # the next argument is evaluated, and the allocator hands the same block to the next temporary (found by tests/test_strided_operands.py)
    d_rgb, d_depth, d_acc = _f32c(d_rgb), _f32c(d_depth), _f32c(d_acc)
    _need_gpu(sigmas, rgbs, z_vals, d_rgb, d_depth, d_acc)
    S = rgbs.shape[-2]
    n_rays = rgbs.numel() // (3 * S) if rgbs.numel() else 0
    dev = rgbs.device
    d_sig = torch.empty(n_rays, S, device=dev)
    d_rgbs = torch.empty(n_rays, S, 3, device=dev)
    d_z = torch.empty(n_rays, S, device=dev) if need_dz else None
    with torch.cuda.device(dev):
        check(_lib.lib().snr_composite_bwd(_p(sigmas), _p(rgbs), _p(z_vals), z_mode, WHITE_BKGD if white_bkgd else 0, n_rays,
                                           rays_per_obj, S, _p(d_rgb), _p(d_depth), _p(d_acc),
                                           _p(d_sig), _p(d_rgbs), _p(d_z), _stream(dev)), "snr_composite_bwd")
    return d_sig, d_rgbs, d_z


class Composite(torch.autograd.Function):
    """Alpha composite of (N,S) densities / (N,S,3) colours; z per z_mode.  Gradients flow to sigmas and
    rgbs always and to z_vals when it is per-ray (family B keeps its box bounds differentiable)."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, z_vals, z_mode, white_bkgd, rays_per_obj):
        sigmas, rgbs, z_vals = _f32c(sigmas), _f32c(rgbs), _f32c(z_vals)
        out = composite_fwd(sigmas, rgbs, z_vals, z_mode, white_bkgd, rays_per_obj)
        ctx.save_for_backward(sigmas, rgbs, z_vals)
        ctx.cfg = (z_mode, white_bkgd, rays_per_obj)
        return out

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_acc):
        sigmas, rgbs, z_vals = ctx.saved_tensors
        z_mode, white, rpo = ctx.cfg
        need_dz = ctx.needs_input_grad[2]
        if need_dz and z_mode != Z_PER_RAY:
            raise SnrError("gradient wrt shared / per-object z_vals is not provided (the reference detaches them, "
                           "src/utils.py:468-469)")
        d_sig, d_rgbs, d_z = composite_bwd(sigmas, rgbs, z_vals, z_mode, white, rpo, d_rgb, d_depth, d_acc, need_dz)
        return d_sig.view(sigmas.shape), d_rgbs.view(rgbs.shape), (d_z.view(z_vals.shape) if need_dz else None), None, None, None


# ------------------------------------------------------------------------------------ loss / metric tail
def loss_tail_fwd(rgb, acc, rgb_tgt, occ, loss_occ_coef, rays_per_obj):
    """(B,4) = [loss, loss_rgb, loss_occ, mse_fg] per object (src/optimizer_nuscenes.py:729-744), one launch."""
    rgb, acc, rgb_tgt, occ = _f32c(rgb), _f32c(acc), _f32c(rgb_tgt), _f32c(occ)
    _need_gpu(rgb, acc, rgb_tgt, occ)
    n = acc.numel()
    if rgb.numel() != 3 * n or rgb_tgt.numel() != 3 * n or occ.numel() != n or rays_per_obj < 1 or n % rays_per_obj:
        raise SnrError(f"loss_tail: rgb {tuple(rgb.shape)}, acc {tuple(acc.shape)}, rgb_tgt {tuple(rgb_tgt.shape)}, occ {tuple(occ.shape)} "
                       f"do not describe whole objects of {rays_per_obj} rays")
    dev = rgb.device
    out = torch.empty(n // rays_per_obj, 4, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_loss_tail_fwd(_p(rgb), _p(acc), _p(rgb_tgt), _p(occ), n, rays_per_obj, float(loss_occ_coef), _p(out), _stream(dev)),
              "snr_loss_tail_fwd")
    return out


class LossTail(torch.autograd.Function):
    """loss (B,) [differentiable wrt rgb and acc_trans] and metrics (B,3) = [loss_rgb, loss_occ, mse_fg] [not differentiable] of the
    optimise iteration.  Backward is one launch that writes the seeds of the render backward."""

    @staticmethod
    def forward(ctx, rgb, acc, rgb_tgt, occ, loss_occ_coef, rays_per_obj):
        rgb, acc, rgb_tgt, occ = _f32c(rgb), _f32c(acc), _f32c(rgb_tgt), _f32c(occ)
        ctx.set_materialize_grads(False)        # (no zero tensors for outputs nobody differentiated: each would be a fill launch)
        out = loss_tail_fwd(rgb, acc, rgb_tgt, occ, loss_occ_coef, rays_per_obj)
        ctx.save_for_backward(rgb, acc, rgb_tgt, occ)
        ctx.cfg = (float(loss_occ_coef), int(rays_per_obj))
        loss, metrics = out[:, 0].contiguous(), out[:, 1:].contiguous()
        ctx.mark_non_differentiable(metrics)
        return loss, metrics

    @staticmethod
    def backward(ctx, g_loss, _g_metrics):
        rgb, acc, rgb_tgt, occ = ctx.saved_tensors
        coef, rpo = ctx.cfg
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            raise SnrError("loss_tail: the targets and occupancy labels are data, no gradient is provided")
        dev = rgb.device
        if g_loss is None:
            return None, None, None, None, None, None
        d_rgb = torch.empty_like(rgb) if ctx.needs_input_grad[0] else None
        d_acc = torch.empty_like(acc) if ctx.needs_input_grad[1] else None
        if d_rgb is not None or d_acc is not None:
            g_loss = _f32c(g_loss)
            with torch.cuda.device(dev):
                check(_lib.lib().snr_loss_tail_bwd(_p(rgb), _p(acc), _p(rgb_tgt), _p(occ), acc.numel(), rpo, coef, _p(g_loss),
                                                   _p(d_rgb), _p(d_acc), _stream(dev)), "snr_loss_tail_bwd")
        return d_rgb, d_acc, None, None, None, None


# ------------------------------------------------------------------------------------ the rest of an optimise iteration
class PoseRays(torch.autograd.Function):
    """Pose parameters -> camera-in-object pose, rays of the pixel grid, stratified depths, in one launch for B objects
    (src/optimizer_nuscenes.py:685-699 + get_rays + the sphere bounds + sample_from_rays' depth vector); backward in one launch.
    rot_vec, trans_vec (B,3); cam_dirs (B,n,3); half_diag (B,); jitter (B,S) or None
    -> cam2opt (B,3,4), rays_o (B*n,3), viewdir (B*n,3), z_vals (B,S) [detached from the pose like the reference's]."""

    @staticmethod
    def forward(ctx, rot_vec, trans_vec, cam_dirs, half_diag, jitter, n_samples, opt_cam_pose):
        rot_vec, trans_vec, cam_dirs, half_diag, jitter = [_f32c(t) for t in (rot_vec, trans_vec, cam_dirs, half_diag, jitter)]
        _need_gpu(rot_vec, trans_vec, cam_dirs, half_diag, jitter)
        B, n = cam_dirs.shape[0], cam_dirs.shape[1]
        if rot_vec.shape != (B, 3) or trans_vec.shape != (B, 3) or half_diag.numel() != B or (jitter is not None and jitter.shape != (B, n_samples)):
            raise SnrError("pose_rays: expected rot_vec / trans_vec (B,3), cam_dirs (B,n,3), half_diag (B,), jitter (B,S)")
        dev = cam_dirs.device
        ctx.set_materialize_grads(False)        # (the kernel takes null for the gradients of outputs that were not used)
        cam2opt = torch.empty(B, 3, 4, device=dev)
        rays_o, viewdir = torch.empty(B * n, 3, device=dev), torch.empty(B * n, 3, device=dev)
        z = torch.empty(B, n_samples, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().snr_pose_rays_fwd(_p(rot_vec), _p(trans_vec), _p(cam_dirs), _p(half_diag), _p(jitter), B, n, int(n_samples),
                                               int(bool(opt_cam_pose)), _p(cam2opt), _p(rays_o), _p(viewdir), _p(z), _stream(dev)), "snr_pose_rays_fwd")
        ctx.save_for_backward(rot_vec, trans_vec, cam_dirs)
        ctx.opt_cam_pose = int(bool(opt_cam_pose))
        ctx.mark_non_differentiable(z)
        return cam2opt, rays_o, viewdir, z

    @staticmethod
    def backward(ctx, d_cam2opt, d_rays_o, d_viewdir, _d_z):
        rot_vec, trans_vec, cam_dirs = ctx.saved_tensors
        if ctx.needs_input_grad[2]:
            raise SnrError("pose_rays: the pixel direction table is data, no gradient is provided")
        B, n = cam_dirs.shape[0], cam_dirs.shape[1]
        dev = cam_dirs.device
        d_rot, d_tr = torch.empty_like(rot_vec), torch.empty_like(trans_vec)
        d_rays_o, d_viewdir, d_cam2opt = _f32c(d_rays_o), _f32c(d_viewdir), _f32c(d_cam2opt)
        with torch.cuda.device(dev):
            check(_lib.lib().snr_pose_rays_bwd(_p(rot_vec), _p(trans_vec), _p(cam_dirs), B, n, ctx.opt_cam_pose, _p(d_rays_o),
                                               _p(d_viewdir), _p(d_cam2opt), _p(d_rot), _p(d_tr), _stream(dev)), "snr_pose_rays_bwd")
        return d_rot, d_tr, None, None, None, None, None


class CamRays(torch.autograd.Function):
    """Camera pose -> rays of a pixel table (+ the stratified depth vector), one launch; backward one launch.  What the public
    ``get_rays`` / ``render_rays_v2`` need per call (src/utils.py:107-135,159-164,468-469) when the pose lives on the GPU.
    c2w (B,3,4); cam_dirs (B,n,3) = [(px-cx)/fx, (py-cy)/fy, 1]; half_diag (B,) or None; jitter (B,S) or None
    -> rays_o (B*n,3), viewdir (B*n,3), z_vals (B,S) [None when half_diag is None; detached from the pose like the reference's]."""

    @staticmethod
    def forward(ctx, c2w, cam_dirs, half_diag, jitter, n_samples):
        c2w, cam_dirs, half_diag, jitter = [_f32c(t) for t in (c2w, cam_dirs, half_diag, jitter)]
        _need_gpu(c2w, cam_dirs, half_diag, jitter)
        B, n = cam_dirs.shape[0], cam_dirs.shape[1]
        if c2w.shape != (B, 3, 4) or cam_dirs.shape[-1] != 3 or (half_diag is not None and half_diag.numel() != B) or \
                (jitter is not None and jitter.shape != (B, n_samples)):
            raise SnrError("cam_rays: expected c2w (B,3,4), cam_dirs (B,n,3), half_diag (B,), jitter (B,S)")
        dev = cam_dirs.device
        ctx.set_materialize_grads(False)
        rays_o, viewdir = torch.empty(B * n, 3, device=dev), torch.empty(B * n, 3, device=dev)
        z = torch.empty(B, n_samples, device=dev) if half_diag is not None else None
        with torch.cuda.device(dev):
            check(_lib.lib().snr_cam_rays_fwd(_p(c2w), _p(cam_dirs), _p(half_diag), _p(jitter), B, n, int(n_samples), _p(rays_o), _p(viewdir), _p(z),
                                              _stream(dev)), "snr_cam_rays_fwd")
        ctx.save_for_backward(c2w, cam_dirs)
        if z is not None:
            ctx.mark_non_differentiable(z)
        return rays_o, viewdir, z

    @staticmethod
    def backward(ctx, d_rays_o, d_viewdir, _d_z):
        c2w, cam_dirs = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise SnrError("cam_rays: the pixel direction table is data, no gradient is provided")
        if d_rays_o is None and d_viewdir is None:
            return None, None, None, None, None
        B, n = cam_dirs.shape[0], cam_dirs.shape[1]
        dev = cam_dirs.device
        d_rays_o, d_viewdir = _f32c(d_rays_o), _f32c(d_viewdir)
        d_c2w = torch.empty_like(c2w)
        with torch.cuda.device(dev):
            check(_lib.lib().snr_cam_rays_bwd(_p(c2w), _p(cam_dirs), B, n, _p(d_rays_o), _p(d_viewdir), _p(d_c2w), _stream(dev)), "snr_cam_rays_bwd")
        return d_c2w, None, None, None, None


def metric_row(loss_out, depth_pred, depth0, first, cam2opt, gt_R, gt_T, opt_cam_pose, row, lidar_count=None):
    """row (B,4) <- [PSNR, depth error over each object's first ``lidar_count`` pixels, rotation error, translation error]
    (src/optimizer_nuscenes.py:739-765); depth0 = the measured depths (first = False) or a buffer that receives the rendered depths of
    the first iteration (first = True)."""
    _need_gpu(loss_out, depth_pred, depth0, cam2opt, gt_R, gt_T, row, lidar_count)
    B = cam2opt.shape[0]
    n_lidar = depth_pred.shape[-1] if depth_pred is not None else 0
    for t in (loss_out, depth_pred, depth0, cam2opt, gt_R, gt_T, row):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise SnrError("metric_row: fp32 contiguous tensors expected")
    if lidar_count is not None and (lidar_count.dtype != torch.int32 or lidar_count.numel() != B or not lidar_count.is_contiguous()):
        raise SnrError("metric_row: lidar_count must be a contiguous int32 (B,) tensor")
    if n_lidar and (depth_pred.numel() != B * n_lidar or depth0 is None or depth0.numel() != B * n_lidar):
        raise SnrError("metric_row: depth_pred / depth0 must both be (B, n_lidar)")
    dev = cam2opt.device
    with torch.cuda.device(dev):
        check(_lib.lib().snr_metric_row(_p(loss_out), _p(depth_pred), _p(depth0), int(n_lidar), int(bool(first)), _p(cam2opt), _p(gt_R), _p(gt_T),
                                        B, int(bool(opt_cam_pose)), _p(row), C.c_void_p(0 if lidar_count is None else lidar_count.data_ptr()),
                                        _stream(dev)), "snr_metric_row")
    return row


class DeviceAdamW:
    """torch.optim.AdamW's update (amsgrad off, same defaults) for up to 4 parameter groups as ONE launch per step
    (src/optimizer_nuscenes.py:1762-1769).  ``groups``: [(parameter tensor, lr), ...]; gradients are read from ``.grad``."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if not 1 <= len(groups) <= 4:
            raise SnrError("DeviceAdamW: 1 to 4 parameter groups")
        self.params = [p for p, _ in groups]
        self.lr = [float(lr) for _, lr in groups]
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise SnrError("DeviceAdamW: fp32 contiguous parameters on the GPU")
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.steps = 0

    def restart(self, scale: float = 1.0):
        """What re-creating the optimiser does (src/optimizer_nuscenes.py:1771-1775): fresh moments and step count, rates times ``scale``."""
        self.lr = [v * scale for v in self.lr]
        for t in self.exp_avg + self.exp_avg_sq:
            t.zero_()
        self.steps = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        n = len(self.params)
        grads = []
        for p in self.params:
            if p.grad is None:
                raise SnrError("DeviceAdamW.step: a parameter has no gradient")
            grads.append(_f32c(p.grad))
        self.steps += 1
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        numel = (C.c_int64 * n)(*[p.numel() for p in self.params])
        lr = (C.c_float * n)(*self.lr)
        dev = self.params[0].device
        with torch.cuda.device(dev):
            check(_lib.lib().snr_adamw_step(arr([p.data for p in self.params]), arr(grads), arr(self.exp_avg), arr(self.exp_avg_sq), numel, lr, n, self.steps,
                                            self.betas[0], self.betas[1], self.eps, self.weight_decay, _stream(dev)), "snr_adamw_step")


class LatentLayers(torch.autograd.Function):
    """The per-object latent layers of a FROZEN decoder (src/model_supnerf.py:253,261) and the biases their outputs fold into, one launch;
    backward to the two codes one launch.  shapecode, texturecode (B,256); w_lat, b_lat, w_nxt, b_nxt: ``model._stacked()``.
    -> z (B, n_lat, 256), latent_bias (B, n_lat, 256) [no gradient: the render backward returns the gradient of z itself]."""

    @staticmethod
    def forward(ctx, shapecode, texturecode, w_lat, b_lat, w_nxt, b_nxt, shape_blocks, texture_blocks):
        shapecode, texturecode = _f32c(shapecode), _f32c(texturecode)
        _need_gpu(shapecode, texturecode, w_lat, b_lat, w_nxt, b_nxt)
        n_lat, B, dev = shape_blocks + texture_blocks, shapecode.shape[0], shapecode.device
        if shapecode.shape != (B, 256) or texturecode.shape != (B, 256) or w_lat.shape != (512, n_lat * 256) or w_nxt.shape != (n_lat * 256, n_lat * 256) \
                or b_lat.numel() != n_lat * 256 or b_nxt.numel() != n_lat * 256:
            raise SnrError("latent_layers: expected codes (B,256), w_lat (512, n_lat*256), w_nxt (n_lat*256, n_lat*256), biases (n_lat*256)")
        z, lb = torch.empty(B, n_lat, 256, device=dev), torch.empty(B, n_lat, 256, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().snr_latent_fwd(_p(shapecode), _p(texturecode), _p(w_lat), _p(b_lat), _p(w_nxt), _p(b_nxt), B, int(shape_blocks),
                                            int(texture_blocks), _p(z), _p(lb), _stream(dev)), "snr_latent_fwd")
        ctx.save_for_backward(z, w_lat)
        ctx.blocks = (int(shape_blocks), int(texture_blocks))
        ctx.mark_non_differentiable(lb)
        ctx.set_materialize_grads(False)
        return z, lb

    @staticmethod
    def backward(ctx, d_z, _d_lb):
        z, w_lat = ctx.saved_tensors
        if d_z is None:
            return (None,) * 8
        if any(ctx.needs_input_grad[2:6]):
            raise SnrError("latent_layers: the stacked weights are constants here (training mode runs the per-layer torch form)")
        sb, tb = ctx.blocks
        B, dev = z.shape[0], z.device
        d_z = _f32c(d_z)
        d_sc = torch.empty(B, 256, device=dev) if ctx.needs_input_grad[0] else None
        d_tc = torch.empty(B, 256, device=dev) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(dev):
            check(_lib.lib().snr_latent_bwd(_p(d_z), _p(z), _p(w_lat), B, sb, tb, _p(d_sc), _p(d_tc), _stream(dev)), "snr_latent_bwd")
        return d_sc, d_tc, None, None, None, None, None, None


class TableAdamW:
    """torch.optim.AdamW's update (amsgrad off, same defaults) for ANY number of tensors as ONE launch per step: the training step's
    optimiser (src/trainer_unified_nuscenes.py:414-422; torch's foreach form is ~18 launches over the 36 tensors, its ``fused=True`` form
    does not reproduce the update on this ROCm build).  ``groups``: [(list of parameters, lr), ...] (up to 4).  Gradients are read from
    the ``.grad`` tensors the parameters hold WHEN THE OPTIMISER IS BUILT (the trainer's bucket views: fixed addresses), which are
    checked again at every step.  After the launch the parameters' version counters are bumped, so caches keyed on them (the packed
    weight stream) see the change."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if not 1 <= len(groups) <= 4:
            raise SnrError("TableAdamW: 1 to 4 parameter groups")
        self.lr = [float(lr) for _, lr in groups]
        self.params, self.group_of = [], []
        for gi, (ps, _) in enumerate(groups):
            for p in ps:
                if p.requires_grad:
                    self.params.append(p); self.group_of.append(gi)
        if not self.params:
            raise SnrError("TableAdamW: no trainable parameter")
        dev = self.params[0].device
        for p in self.params:
            if not p.is_cuda or p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise SnrError("TableAdamW: fp32 contiguous parameters on one GPU")
            if p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.shape != p.shape or p.grad.device != dev:
                raise SnrError("TableAdamW: every parameter needs its dense fp32 .grad buffer before the optimiser is built (GradBucket)")
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.steps = 0
        self._key = None
        self._table = None
        self._build_table()

    def _build_table(self):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in self.params)
        if key == self._key:
            return
        rows = [[p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), g]
                for p, m, v, g in zip(self.params, self.exp_avg, self.exp_avg_sq, self.group_of)]
        self._table = torch.tensor(rows, dtype=torch.int64).to(self.params[0].device)
        self._max_numel = max(p.numel() for p in self.params)
        self._key = key

    def zero_grad(self, set_to_none=False):
        if set_to_none:
            raise SnrError("TableAdamW reads the gradients from fixed buffers: clear them in place (GradBucket.zero)")
        for p in self.params:
            p.grad.zero_()

    def step(self):
        for p in self.params:
            if p.grad is None:
                raise SnrError("TableAdamW.step: a parameter lost its .grad buffer (zero_grad(set_to_none=True)?)")
        self._build_table()          # (a moved parameter or a replaced gradient buffer: new table)
        self.steps += 1
        lr = (C.c_float * len(self.lr))(*self.lr)
        dev = self.params[0].device
        with torch.cuda.device(dev):
            check(_lib.lib().snr_adamw_table_step(_p(self._table), len(self.params), int(self._max_numel), lr, len(self.lr), self.steps,
                                                  self.betas[0], self.betas[1], self.eps, self.weight_decay, _stream(dev)), "snr_adamw_table_step")
        torch.autograd.graph.increment_version(self.params)


# ------------------------------------------------------------------------------------ decoder on points
def decoder_fwd(xyz, viewdir, latent, packed, shape_blocks, texture_blocks, save_masks=False, precision="fp32", activations=None):
    """xyz, viewdir (P,3); latent (B,NLAT,256) -> sigmas (P,), rgbs (P,3)[, relu masks]."""
    xyz, viewdir, latent = _f32c(xyz), _f32c(viewdir), _f32c(latent)
    _need_gpu(xyz, viewdir, latent, packed)
    P = xyz.shape[0]
    B = latent.shape[0]
    if P % B:
        raise SnrError("number of points is not divisible by the number of objects")
    dev = xyz.device
    sig = torch.empty(P, device=dev)
    rgb = torch.empty(P, 3, device=dev)
    masks = None
    if save_masks:
        masks = torch.empty(_lib.lib().snr_mask_bytes(P, shape_blocks, texture_blocks), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_decoder_fwd(_p(xyz), _p(viewdir), _p(latent), _p(packed), P, P // B if B else 1, shape_blocks,
                                         texture_blocks, _p(sig), _p(rgb), _p(masks), _p(activations),
                                         resolve_precision(precision, shape_blocks, texture_blocks, P // B if B else 1), _stream(dev)),
              "snr_decoder_fwd")
    return sig, rgb, masks


def decoder_bwd(xyz, viewdir, latent, packed, masks, sigmas, d_sig, d_rgb, shape_blocks, texture_blocks,
                need_latent=True, need_xyz=True, need_dir=True, precision="fp32", layer_grads=None):
    xyz, viewdir, latent, sigmas = _f32c(xyz), _f32c(viewdir), _f32c(latent), _f32c(sigmas)
    _need_gpu(xyz, viewdir, latent, packed, masks, sigmas, d_sig, d_rgb, layer_grads)
    if masks is None or sigmas is None:
        raise SnrError("decoder_bwd needs the ReLU bits and densities saved by decoder_fwd(..., save_masks=True)")
    masks, d_sig, d_rgb = masks.contiguous(), _f32c(d_sig), _f32c(d_rgb)
    P, B = xyz.shape[0], latent.shape[0]
    dev = xyz.device
    if need_latent and shape_blocks + texture_blocks > 0 and (P // B) % 32:
        raise SnrError(f"gradient wrt the latent codes needs whole 32-point wave tiles per object, got {P // B} points per object "
                       f"({P} points / {B} codes): pad the ray batch of every object to a multiple of 32 / n_samples rays, or detach the codes")
    d_latent = torch.empty_like(latent) if need_latent else None
    d_xyz = torch.empty_like(xyz) if need_xyz else None
    d_dir = torch.empty_like(viewdir) if need_dir else None
    ws_bytes = _lib.lib().snr_decoder_bwd_ws_bytes(P, P // B, shape_blocks, texture_blocks)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_decoder_bwd(_p(xyz), _p(viewdir), _p(latent), _p(packed), _p(masks), _p(sigmas), _p(d_sig),
                                         _p(d_rgb), P, P // B, shape_blocks, texture_blocks, _p(d_latent), _p(d_xyz),
                                         _p(d_dir), _p(layer_grads), _p(ws), ws_bytes, resolve_precision(precision, shape_blocks, texture_blocks, P // B),
                                         _stream(dev)), "snr_decoder_bwd")
    return d_latent, d_xyz, d_dir


def _tile_pad(per_obj: int, unit: int = 1) -> int:
    """Items per object after padding so that items x unit is a multiple of the 32-point wave tile (0 = no padding needed)."""
    if (per_obj * unit) % 32 == 0:
        return 0
    step = 32 // math.gcd(32, unit)
    return -(-per_obj // step) * step


def _pad_rows(t, B, n, n_pad):
    """(B*n, ...) -> (B*n_pad, ...): every object's block padded with zero rows (dummy points / rays: their upstream gradients are zero)."""
    if t is None:
        return None
    v = t.reshape(B, n, *t.shape[1:])
    out = v.new_zeros(B, n_pad, *t.shape[1:])
    out[:, :n] = v
    return out.reshape(B * n_pad, *t.shape[1:])


def _unpad_rows(t, B, n, n_pad):
    if t is None:
        return None
    return t.reshape(B, n_pad, *t.shape[1:])[:, :n].reshape(B * n, *t.shape[1:])


class DecoderPoints(torch.autograd.Function):
    """SUPNeRF.forward on explicit points.  Differentiable wrt latent terms, xyz and viewdir (weights are
    treated as constants on this path)."""

    @staticmethod
    def forward(ctx, xyz, viewdir, latent, packed, shape_blocks, texture_blocks, precision="fp32"):
        xyz, viewdir, latent = _f32c(xyz), _f32c(viewdir), _f32c(latent)
        need = any(ctx.needs_input_grad[:3])
        B, P = max(latent.shape[0], 1), xyz.shape[0]
        # the latent-gradient reduction works on whole 32-point wave tiles per object: other point counts (the reference takes any) are padded
        # per object with dummy points whose upstream gradients are zero
        n_pad = _tile_pad(P // B) if (ctx.needs_input_grad[2] and shape_blocks + texture_blocks > 0 and P % B == 0 and P > 0) else 0
        ctx.pad = (B, P // B, n_pad)
        if n_pad:
            xyz, viewdir = _pad_rows(xyz, B, P // B, n_pad), _pad_rows(viewdir, B, P // B, n_pad)
        prec = resolve_precision(precision, shape_blocks, texture_blocks, xyz.shape[0] // B)
        sig, rgb, masks = decoder_fwd(xyz, viewdir, latent, packed, shape_blocks, texture_blocks, save_masks=need, precision=prec)
        if need:
            ctx.save_for_backward(xyz, viewdir, latent, packed, masks, sig)
            ctx.cfg = (shape_blocks, texture_blocks, resolve_precision(precision, shape_blocks, texture_blocks, xyz.shape[0] // B, backward=True))
        if n_pad:
            return _unpad_rows(sig, B, P // B, n_pad), _unpad_rows(rgb, B, P // B, n_pad)
        return sig, rgb

    @staticmethod
    def backward(ctx, d_sig, d_rgb):
        xyz, viewdir, latent, packed, masks, sig = ctx.saved_tensors
        sb, tb, prec = ctx.cfg
        B, n, n_pad = ctx.pad
        if n_pad:
            d_sig, d_rgb = _pad_rows(_f32c(d_sig), B, n, n_pad), _pad_rows(_f32c(d_rgb), B, n, n_pad)
        d_lat, d_xyz, d_dir = decoder_bwd(xyz, viewdir, latent, packed, masks, sig, d_sig, d_rgb, sb, tb,
                                          ctx.needs_input_grad[2], ctx.needs_input_grad[0], ctx.needs_input_grad[1], precision=prec)
        if n_pad:
            d_xyz, d_dir = _unpad_rows(d_xyz, B, n, n_pad), _unpad_rows(d_dir, B, n, n_pad)
        return d_xyz, d_dir, d_lat, None, None, None, None


def weight_grad(G, n_out, X, n_in, want_bias=True, out=None, ws=None, precision="fp32"):
    """dW (n_out, n_in) = G[:, :n_out]^T X[:, :n_in] and db (n_out,) = column sums of G, one split-K MFMA launch + one reduction
    (include/supnerf_hip.h: snr_weight_grad).  G, X: 2-D fp32 row-major views (a column slice of a wider buffer is fine).  ``out``:
    optional (dW_view, db) to write into (dW_view may be a column block of a wider matrix).  ``precision``: "fp32" (exact) or "bf16x3"
    (split-bf16 products, ~2^-17 operand error, several times faster; bias sums and narrow heads stay fp32)."""
    if precision not in PRECISIONS:
        raise SnrError(f"weight_grad: precision must be 'fp32' or 'bf16x3', got {precision!r}")
    _need_gpu(G, X)
    if G.dim() != 2 or X.dim() != 2 or G.shape[0] != X.shape[0] or G.stride(1) != 1 or X.stride(1) != 1 or G.dtype != torch.float32 or X.dtype != torch.float32:
        raise SnrError(f"weight_grad: G {tuple(G.shape)} / X {tuple(X.shape)} must be fp32 (P, n) row-major views of the same points")
    if n_out > G.shape[1] or n_in > X.shape[1]:
        raise SnrError("weight_grad: n_out / n_in exceed the operands' columns")
    P, dev = G.shape[0], G.device
    dW, db = out if out is not None else (torch.empty(n_out, n_in, device=dev), torch.empty(n_out, device=dev) if want_bias else None)
    if dW.stride(1) != 1:
        raise SnrError("weight_grad: dW must be row-major")
    nbytes = _lib.lib().snr_weight_grad_ws_bytes(P, n_out, n_in)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_weight_grad(_p_rows(G), G.stride(0), n_out, _p_rows(X), X.stride(0), n_in, P, _p_rows(dW), dW.stride(0), _p(db), PRECISIONS[precision],
                                         _p(ws), ws.numel(), _stream(dev)), "snr_weight_grad")
    return dW, db


def pe_points(xyz, viewdir):
    """(P,96) = PE(xyz) (63 columns, 1 zero) | PE(viewdir) (27 columns, 5 zeros): include/supnerf_hip.h snr_pe_points."""
    xyz, viewdir = _f32c(xyz), _f32c(viewdir)
    _need_gpu(xyz, viewdir)
    P, dev = xyz.shape[0], xyz.device
    if xyz.shape != (P, 3) or viewdir.shape != (P, 3):
        raise SnrError("pe_points: xyz and viewdir must both be (P,3)")
    out = torch.empty(P, 96, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_pe_points(_p(xyz), _p(viewdir), P, _p(out), _stream(dev)), "snr_pe_points")
    return out


_PACK_CACHE = {}
_PACK_LOCK = threading.Lock()


def _packed_for(names, weights, shape_blocks, texture_blocks):
    """The packed stream of these per-point weights, re-packed only when a tensor changed (every optimiser step bumps the versions; an
    evaluation pass between two steps, or a second forward before the update, re-uses the buffer).  One slot per device (DataParallel enters
    from one thread per GPU).  A hit must PROVE identity: the entry holds weak references to the very tensors it was packed from, and
    ``ref() is w`` is required for every one -- (data_ptr, _version) alone would also match a NEW model whose weights the caching
    allocator placed at a freed model's addresses (same construction, same version counters), and the step would silently run on the
    old model's weights."""
    dev = (weights[0].device, raw_stream(weights[0].device) if weights[0].is_cuda else 0)   # (one slot per device AND stream)
    key = (shape_blocks, texture_blocks) + tuple((w.data_ptr(), w._version) for w in weights)
    with _PACK_LOCK:
        hit = _PACK_CACHE.get(dev)
        if hit is not None and hit[0] == key and len(hit[1]) == len(weights) and all(r() is w for r, w in zip(hit[1], weights)):
            return hit[2]
    packed = pack_weights(dict(zip(names, weights)), shape_blocks, texture_blocks)
    with _PACK_LOCK:
        while len(_PACK_CACHE) >= 16:
            _PACK_CACHE.pop(next(iter(_PACK_CACHE)))
        _PACK_CACHE[dev] = (key, [weakref.ref(w) for w in weights], packed)
    return packed


class DecoderPointsTrain(torch.autograd.Function):
    """Training-mode decoder (SURVEY 8a9 mode B): like DecoderPoints but the per-point decoder WEIGHTS are inputs too and
    receive gradients.  The layer-chain kernels additionally write every layer's input X_l (forward) and pre-activation
    gradient G_l (backward) to HBM, and the weight gradients dW_l = G_l^T X_l, db_l = sum_p G_l come from the split-K MFMA kernels
    behind ``weight_grad`` (no library BLAS).  ``precision``: "fp32" = exact fp32 MFMA throughout; "bf16x3" = split-bf16
    products in all three (the chains and the weight-gradient product; bias sums and the two narrow heads stay fp32); a pair
    (forward chain, backward chain) -- the products follow the backward chain -- or a triple (forward chain, backward chain, products),
    e.g. ("fp32", "bf16x3", "bf16x3"): exact fp32 forward, split-bf16 backward chain (on the ReLU bits the forward saved) and
    weight-gradient products.  ``weights`` = the
    per-point tensors in per_point_tensor_names order."""

    @staticmethod
    def forward(ctx, xyz, viewdir, latent, shape_blocks, texture_blocks, precision, *weights):
        xyz, viewdir, latent = _f32c(xyz), _f32c(viewdir), _f32c(latent)
        # ragged point counts per object: padded to whole wave tiles with dummy points like DecoderPoints (zero upstream gradient, so they add
        # nothing to any weight gradient either)
        B, P0 = max(latent.shape[0], 1), xyz.shape[0]
        n_pad = _tile_pad(P0 // B) if (shape_blocks + texture_blocks > 0 and P0 % B == 0 and P0 > 0) else 0
        ctx.pad = (B, P0 // B, n_pad)
        if n_pad:
            xyz, viewdir = _pad_rows(xyz, B, P0 // B, n_pad), _pad_rows(viewdir, B, P0 // B, n_pad)
        # one arithmetic for the whole step: the forward / backward layer chains and the weight-gradient products
        # ``precision``: one name for the whole step, (forward chain, backward chain) -- the products follow the backward chain -- or (forward
        # chain, backward chain, products); ("fp32", "auto", "bf16x3") is what "auto" trains in (model.forward): the forward chain decides
        # where a training run ends up, what runs behind it does not
        wgrad_override = precision[2] if isinstance(precision, tuple) and len(precision) == 3 else None
        per_obj = xyz.shape[0] // max(latent.shape[0], 1)
        prec = resolve_precision(precision, shape_blocks, texture_blocks, per_obj)
        prec_bwd = resolve_precision(precision, shape_blocks, texture_blocks, per_obj, backward=True)
        wgrad_precision = wgrad_override or ("bf16x3" if prec_bwd == BF16X3 else "fp32")       # (a pair: the products follow the backward chain)
        if wgrad_precision not in ("fp32", "bf16x3"):
            raise SnrError(f"weight-gradient products run in 'fp32' or 'bf16x3', not {wgrad_precision!r}")
        names = per_point_tensor_names(shape_blocks, texture_blocks)
        packed = _packed_for(names, weights, shape_blocks, texture_blocks)
        P, dev = xyz.shape[0], xyz.device
        n_slots = shape_blocks + texture_blocks + 4
        act = torch.empty(n_slots, P, 256, device=dev)
        sig, rgb, masks = decoder_fwd(xyz, viewdir, latent, packed, shape_blocks, texture_blocks, save_masks=True, precision=prec,
                                      activations=act)
        ctx.save_for_backward(xyz, viewdir, latent, packed, masks, sig, act, *weights)
        ctx.cfg = (shape_blocks, texture_blocks, wgrad_precision, prec_bwd)
        if n_pad:
            return _unpad_rows(sig, B, P0 // B, n_pad), _unpad_rows(rgb, B, P0 // B, n_pad)
        return sig, rgb

    @staticmethod
    def backward(ctx, d_sig, d_rgb):
        xyz, viewdir, latent, packed, masks, sig, act, *weights = ctx.saved_tensors
        sb, tb, wprec, prec = ctx.cfg
        P, dev = xyz.shape[0], xyz.device
        n_slots = sb + tb + 4
        G = torch.empty(n_slots, P, 256, device=dev)
        d_sig, d_rgb = _f32c(d_sig), _f32c(d_rgb)
        B_, n_, n_pad = ctx.pad
        if n_pad:
            d_sig, d_rgb = _pad_rows(d_sig, B_, n_, n_pad), _pad_rows(d_rgb, B_, n_, n_pad)
        d_lat, d_xyz, d_dir = decoder_bwd(xyz, viewdir, latent, packed, masks, sig, d_sig, d_rgb, sb, tb,
                                          ctx.needs_input_grad[2], ctx.needs_input_grad[0], ctx.needs_input_grad[1], precision=prec,
                                          layer_grads=G)
        # ---- weight gradients: dW_l = G_l^T X_l, db_l = sum_p G_l on the split-K MFMA kernel (snr_weight_grad), one launch + one
        # reduction per layer; X of layer 0 and the direction features are the positional encodings, one launch of snr_pe_points (``pe``:
        # columns 0..63 PE(xyz), 64..95 PE(dir); round 2 recomputed them with torch.sin / cos / cat: 0.3 ms a step)
        pe = pe_points(xyz, viewdir)
        li_view, li_rgb0 = sb + 2, sb + tb + 3
        ws = torch.empty(_lib.lib().snr_weight_grad_ws_bytes(P, 256, 256), dtype=torch.uint8, device=dev)
        by_layer = {}
        for li in range(n_slots):              # MFMA layers in order; the two small heads follow
            n_out = 128 if li == li_rgb0 else 256
            if li == 0:
                by_layer[li] = weight_grad(G[li], n_out, pe[:, :64], 64, ws=ws, precision=wprec)
                by_layer[li] = (by_layer[li][0][:, :63].contiguous(), by_layer[li][1])
            elif li == li_view:
                dW = torch.empty(256, 256 + 28, device=dev)
                _, db = weight_grad(G[li], 256, act[li - 1], 256, out=(dW[:, :256], torch.empty(256, device=dev)), ws=ws, precision=wprec)
                weight_grad(G[li], 256, pe[:, 64:], 28, out=(dW[:, 256:], None), ws=ws, precision=wprec)
                by_layer[li] = (dW[:, :283].contiguous(), db)
            else:
                by_layer[li] = weight_grad(G[li], n_out, act[li - 1], 256, ws=ws, precision=wprec)
        # sigma head: pre = w . y4 + b with y4 = input of enc_viewdir; d pre = d_sig * sigmoid(pre) = d_sig * (1 - exp(-sigma))
        dpre = (d_sig * (1 - torch.exp(-sig))).reshape(P, 1)
        d_sigma_w, d_sigma_b = weight_grad(dpre, 1, act[li_view - 1], 256, ws=ws)
        d_rgb2_w, d_rgb2_b = weight_grad(d_rgb.reshape(P, 3), 3, act[n_slots - 1], 128, ws=ws)
        out = []
        order = [0] + list(range(1, sb + 1)) + [sb + 1, "sigma", li_view] + list(range(sb + 3, sb + 3 + tb)) + [li_rgb0, "rgb2"]
        for k in order:
            if k == "sigma":
                out += [d_sigma_w, d_sigma_b]
            elif k == "rgb2":
                out += [d_rgb2_w, d_rgb2_b]
            else:
                out += list(by_layer[k])
        if n_pad:
            d_xyz, d_dir = _unpad_rows(d_xyz, B_, n_, n_pad), _unpad_rows(d_dir, B_, n_, n_pad)
        return (d_xyz, d_dir, d_lat, None, None, None, *out)


# ------------------------------------------------------------------------------------ fused render
def _render_args(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, frame, xyz_mul, z_mode, flags, rays_per_obj,
                 n_samples, shape_blocks, texture_blocks, precision=FP32, latent_bias=None, box_half=None, rng=None):
    a = RenderArgs()
    a.rays_o, a.rays_d, a.t_vals = _ptr(rays_o), _ptr(rays_d), _ptr(t_vals)
    a.xyz_div, a.z_scale = _ptr(xyz_div), _ptr(z_scale)
    a.latent, a.packed = _ptr(latent), _ptr(packed)
    a.frame = (C.c_float * 9)(*[float(v) for v in frame])
    a.xyz_mul = float(xyz_mul)
    a.z_mode, a.flags = int(z_mode), int(flags)
    a.n_rays, a.rays_per_obj = int(rays_o.shape[0]), int(rays_per_obj)
    a.n_samples, a.shape_blocks, a.texture_blocks = int(n_samples), int(shape_blocks), int(texture_blocks)
    a.precision = int(precision)
    a.latent_bias = _ptr(latent_bias)
    a.box_half = _ptr(box_half)
    a.rng_seed, a.rng_offset, a.rng_threads = [int(v) & 0xFFFFFFFFFFFFFFFF for v in (rng if rng is not None else (0, 0, 0))]
    # sizes the kernels index with: a short buffer here is an out-of-bounds read on the device
    N, S, B = a.n_rays, a.n_samples, max(a.n_rays // max(a.rays_per_obj, 1), 1)
    if rays_o.numel() != 3 * N or rays_d.numel() != 3 * N:
        raise SnrError(f"render: rays_o {tuple(rays_o.shape)} / rays_d {tuple(rays_d.shape)} are not (N,3)")
    want_t = {Z_SHARED: S, Z_PER_OBJECT: B * S, Z_PER_RAY: N * S, Z_BOX: N * S}[a.z_mode]
    if t_vals is not None and t_vals.numel() != want_t:
        raise SnrError(f"render: depths / jitter hold {t_vals.numel()} values, z_mode {a.z_mode} needs {want_t}")
    if a.z_mode == Z_BOX:
        if box_half is None or box_half.numel() != 3 * B or z_scale is None:
            raise SnrError("render: box sampling needs box_half (B,3) and z_scale (B,)")
    elif t_vals is None or xyz_div is None:
        raise SnrError("render: depths and xyz_div are required")
    for t, name in ((xyz_div, "xyz_div"), (z_scale, "z_scale")):
        if t is not None and t.numel() < B:
            raise SnrError(f"render: {name} holds {t.numel()} values for {B} objects")
    return a


def reserve_rand_like(dev, numel):
    """(seed, offset, threads) describing what ``torch.rand_like`` of ``numel`` floats would draw from ``dev``'s default generator
    right now, and advance the generator exactly as that call would (aten/src/ATen/native/cuda/DistributionTemplates.h:
    calc_execution_policy).  The box-sampling kernels regenerate those numbers themselves (family B's jitter, src/renderer.py:40), so a
    seeded run consumes the generator like the reference without an (N,S) tensor or an extra launch."""
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    props = torch.cuda.get_device_properties(idx)
    block = 256
    grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // block), (numel + block - 1) // block)
    threads = max(grid, 1) * block
    seed, offset = gen.initial_seed(), gen.get_offset()
    gen.set_offset(offset + ((max(numel, 1) - 1) // (threads * 4) + 1) * 4)
    return seed, offset, threads


def render_probe(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, cfg, precision):
    """The fused forward once more, without autograd and WITHOUT consuming the device generator (box sampling with in-kernel jitter draws
    from the state the real launch is about to reserve): (rgb, depth, acc_trans) in ``precision``.  Used by the range guard of ``auto``."""
    c = copy.copy(cfg)
    c.precision = precision
    with torch.no_grad():
        if c.z_mode == Z_BOX and t_vals is None and c.rng is None:
            dev = torch.device(rays_o.device)
            gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
            off = gen.get_offset()
            c.rng = reserve_rand_like(dev, rays_o.shape[0] * c.n_samples)
            gen.set_offset(off)
        return render_fwd(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, c)[:3]


def fused_supported(n_samples: int) -> bool:
    """The single-launch render needs every ray inside one 128-point workgroup tile."""
    return 1 <= n_samples <= 128 and 128 % n_samples == 0


class RenderCfg:
    """Per-launch constants of the render operators (not tensors)."""

    def __init__(self, n_samples, z_mode, rays_per_obj, shape_blocks, texture_blocks, frame=IDENTITY_FRAME, xyz_mul=1.0,
                 white_bkgd=False, metric_z=False, precision=None, box_half=None, box_detach=False):
        self.n_samples, self.z_mode, self.rays_per_obj = n_samples, z_mode, rays_per_obj
        # "fp32" | "bf16x3" | "auto" | None.  None = not chosen here: ``model.fused_render`` substitutes the module's ``precision``; the
        # bare operators (render_fwd, encode) treat it as "auto", the library default.  (It used to default to "fp32", which made
        # every caller that did not pass it run the exact kernels whatever the module said.)
        self.precision = precision
        self.shape_blocks, self.texture_blocks = shape_blocks, texture_blocks
        self.frame, self.xyz_mul = tuple(float(v) for v in frame), float(xyz_mul)
        self.flags = (WHITE_BKGD if white_bkgd else 0) | (METRIC_Z if metric_z else 0) | (BOX_DETACH if box_detach else 0)
        # optional (B, NLAT, 256) fp32 on the device: the latent terms folded into the next layers' biases (model.latent_biases);
        # forward only, no gradient flows through it (the backward kernel returns the gradient of the latent terms themselves)
        self.latent_bias = None
        # z_mode Z_BOX (family B, src/renderer.py:91-115): (B,3) half extents of the objects' boxes on the device; the depths argument of the
        # operators then is the (N,S) jitter or None = drawn in the kernel from ``rng`` = (seed, offset, threads), see reserve_rand_like
        self.box_half = box_half
        self.rng = None


def _box_rng(cfg, t_vals, dev, n_points):
    """Box sampling without a caller-supplied jitter: fix the generator state the kernels will draw from (once per forward; the
    backward replays it from the same cfg)."""
    if cfg.z_mode == Z_BOX and t_vals is None and cfg.rng is None:
        cfg.rng = reserve_rand_like(dev, n_points)


def render_fwd(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, cfg: RenderCfg, save_for_bwd=False):
    rays_o, rays_d, t_vals, xyz_div, z_scale, latent = [_f32c(t) for t in (rays_o, rays_d, t_vals, xyz_div, z_scale, latent)]
    _need_gpu(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed)
    if not fused_supported(cfg.n_samples):
        raise SnrError(f"fused render needs n_samples dividing 128 (got {cfg.n_samples}); use the unfused operators")
    dev = rays_o.device
    N, S = rays_o.shape[0], cfg.n_samples
    rgb = torch.empty(N, 3, device=dev)
    depth = torch.empty(N, device=dev)
    acc = torch.empty(N, device=dev)
    sig = rgbs = masks = None
    if save_for_bwd:
        sig = torch.empty(N * S, device=dev)
        rgbs = torch.empty(N * S, 3, device=dev)
        masks = torch.empty(_lib.lib().snr_mask_bytes(N * S, cfg.shape_blocks, cfg.texture_blocks), dtype=torch.uint8, device=dev)
    prec = resolve_precision(cfg.precision, cfg.shape_blocks, cfg.texture_blocks, cfg.rays_per_obj * S)
    lb = getattr(cfg, "latent_bias", None)
    if lb is not None:
        lb = _f32c(lb.detach())
        if lb.shape != latent.shape or lb.device != dev:
            raise SnrError(f"latent_bias must match the latent terms: {tuple(lb.shape)} on {lb.device} vs {tuple(latent.shape)} on {dev}")
    _box_rng(cfg, t_vals, dev, N * S)
    bh = _f32c(cfg.box_half)
    a = _render_args(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, cfg.frame, cfg.xyz_mul, cfg.z_mode, cfg.flags,
                     cfg.rays_per_obj, S, cfg.shape_blocks, cfg.texture_blocks, prec, latent_bias=lb, box_half=bh, rng=cfg.rng)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_render_fwd(C.byref(a), _p(rgb), _p(depth), _p(acc), _p(sig), _p(rgbs), _p(masks), _stream(dev)),
              "snr_render_fwd")
    return rgb, depth, acc, sig, rgbs, masks


def pad_render_inputs(rays_o, rays_d, t_vals, z_scale, cfg, B, n, n_pad):
    """Every object's n rays padded to n_pad with dummy rays (whole 32-point wave tiles per object); returns the padded operands and a
    copy of ``cfg`` for the padded launch.  Box sampling with in-kernel jitter (Z_BOX, no jitter tensor): the kernel's Philox stream is
    indexed by the LAUNCH's point index, which padding shifts -- so the reference's draw, ``torch.rand_like`` of the caller's (N,S) table
    (src/renderer.py:40: the same numbers and the same generator consumption), is materialised BEFORE padding and padded like any other
    jitter table; seeded parity with the reference survives ragged ray counts."""
    cfg = copy.copy(cfg)
    if cfg.z_mode == Z_BOX and t_vals is None and cfg.rng is None:
        t_vals = torch.rand(B * n, cfg.n_samples, device=rays_o.device)
    rays_o, rays_d = _pad_rows(rays_o, B, n, n_pad), _pad_rows(rays_d, B, n, n_pad)
    if cfg.z_mode in (Z_PER_RAY, Z_BOX) and t_vals is not None:
        t_vals = _pad_rows(t_vals, B, n, n_pad)
    if z_scale is not None and z_scale.numel() == B * n:
        z_scale = _pad_rows(z_scale, B, n, n_pad)
    cfg.rays_per_obj = n_pad
    if cfg.z_mode == Z_BOX:             # (a dummy ray with direction 0 would divide 0 by 0 in the slab test: give it any unit direction)
        rays_d = rays_d.clone()
        rays_d.view(B, n_pad, 3)[:, n:, 2] = 1.0
    return rays_o, rays_d, t_vals, z_scale, cfg


class FusedRender(torch.autograd.Function):
    """rays -> (rgb, depth, acc_trans) in one launch; backward in one launch (+ a small reduction).  ``t_vals``: the depths in
    ``cfg.z_mode``'s layout; for Z_BOX the (N,S) jitter, or None (drawn in the kernel).  Gradients: rays_o, rays_d, latent always; t_vals for
    per-ray depths; for Z_BOX the gradient through the box bounds is part of d_rays_o / d_rays_d."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, cfg):
        rays_o, rays_d, t_vals, xyz_div, z_scale, latent = [_f32c(t) for t in (rays_o, rays_d, t_vals, xyz_div, z_scale, latent)]
        need = any(ctx.needs_input_grad[:6])
        ctx.set_materialize_grads(False)        # (e.g. the depth output of an optimise iteration: the kernel takes null upstream gradients)
        # few samples per ray (S < 32) and a ray count that leaves a partial 32-point wave tile per object: pad every object with dummy rays
        # (see DecoderPoints); the outputs and gradients of the dummies are dropped
        n = cfg.rays_per_obj
        B = rays_o.shape[0] // n if n else 0
        n_pad = _tile_pad(n, cfg.n_samples) if (ctx.needs_input_grad[5] and cfg.shape_blocks + cfg.texture_blocks > 0 and B > 0) else 0
        ctx.pad = (B, n, n_pad)
        cfg = copy.copy(cfg)                    # (the launch's generator state is recorded in it: one copy per call)
        if n_pad:
            rays_o, rays_d, t_vals, z_scale, cfg = pad_render_inputs(rays_o, rays_d, t_vals, z_scale, cfg, B, n, n_pad)
        rgb, depth, acc, sig, rgbs, masks = render_fwd(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, cfg, save_for_bwd=need)
        if need:
            ctx.save_for_backward(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, sig, rgbs, masks)
            ctx.cfg = cfg
        if n_pad:
            return _unpad_rows(rgb, B, n, n_pad), _unpad_rows(depth, B, n, n_pad), _unpad_rows(acc, B, n, n_pad)
        return rgb, depth, acc

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_acc):
        rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, sig, rgbs, masks = ctx.saved_tensors
        cfg = ctx.cfg
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            raise SnrError("xyz_div / z_scale are per-object constants (object size); no gradient is provided")
        need_t = ctx.needs_input_grad[2]
        if need_t and cfg.z_mode != Z_PER_RAY:
            raise SnrError("gradient wrt shared / per-object depths or the box jitter is not provided (the reference detaches them)")
        B, n, n_pad = ctx.pad
        if d_rgb is None and d_depth is None and d_acc is None:
            return None, None, None, None, None, None, None, None
        if n_pad:
            d_rgb, d_depth, d_acc = [_pad_rows(_f32c(t), B, n, n_pad) for t in (d_rgb, d_depth, d_acc)]
        d_o, d_d, d_t, d_lat = render_bwd(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, cfg, sig, rgbs, masks, d_rgb, d_depth, d_acc,
                                          need_o=ctx.needs_input_grad[0], need_d=ctx.needs_input_grad[1], need_t=need_t,
                                          need_latent=ctx.needs_input_grad[5])
        if n_pad:
            d_o, d_d, d_t = _unpad_rows(d_o, B, n, n_pad), _unpad_rows(d_d, B, n, n_pad), _unpad_rows(d_t, B, n, n_pad)
        return d_o, d_d, d_t, None, None, d_lat, None, None


def render_bwd(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, cfg: RenderCfg, sig, rgbs, masks, d_rgb, d_depth, d_acc,
               need_o=True, need_d=True, need_t=False, need_latent=True):
    """Backward of the fused render on what ``render_fwd(..., save_for_bwd=True)`` saved: one launch + the small reduction of the
    latent-term partials.  Returns (d_rays_o, d_rays_d, d_t, d_latent), None where not asked for.  Any operand may be a strided view
    (get_rays' origins, for one, are a stride-0 view of the pose's 3 numbers): everything is made dense here."""
    rays_o, rays_d, t_vals, xyz_div, z_scale, latent = [_f32c(t) for t in (rays_o, rays_d, t_vals, xyz_div, z_scale, latent)]
    sig, rgbs, d_rgb, d_depth, d_acc = [_f32c(t) for t in (sig, rgbs, d_rgb, d_depth, d_acc)]
    _need_gpu(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, sig, rgbs, masks, d_rgb, d_depth, d_acc)
    if sig is None or rgbs is None or masks is None:
        raise SnrError("render_bwd needs what render_fwd(..., save_for_bwd=True) saved: per-point sigmas, rgbs and the ReLU bits")
    if cfg.z_mode == Z_BOX and t_vals is None and cfg.rng is None:
        raise SnrError("render_bwd: box sampling with in-kernel jitter needs the cfg the forward ran with (it holds the generator state)")
    masks, packed = masks.contiguous(), packed.contiguous()
    dev = rays_o.device
    N, S = rays_o.shape[0], cfg.n_samples
    if sig.numel() != N * S or rgbs.numel() != 3 * N * S or masks.numel() < _lib.lib().snr_mask_bytes(N * S, cfg.shape_blocks, cfg.texture_blocks):
        raise SnrError(f"render_bwd: saved sigmas / rgbs / ReLU bits do not belong to {N} rays x {S} samples")
    for t, k, name in ((d_rgb, 3 * N, "d_rgb"), (d_depth, N, "d_depth"), (d_acc, N, "d_acc")):
        if t is not None and t.numel() != k:
            raise SnrError(f"render_bwd: {name} holds {t.numel()} values, expected {k}")
    if need_latent and cfg.shape_blocks + cfg.texture_blocks > 0 and (cfg.rays_per_obj * cfg.n_samples) % 32:
        raise SnrError(f"gradient wrt the latent codes needs whole 32-point wave tiles per object, got {cfg.rays_per_obj} rays x {cfg.n_samples} "
                       "samples per object: pad the ray batch of every object so that rays x samples is a multiple of 32, or detach the codes")
    if need_t and cfg.z_mode != Z_PER_RAY:
        raise SnrError("render_bwd: d_t exists for per-ray depths only")
    d_lat = torch.empty_like(latent) if need_latent else None
    d_o = torch.empty_like(rays_o) if need_o else None      # (the kernel writes every ray's gradient exactly once: no memset)
    d_d = torch.empty_like(rays_d) if need_d else None
    d_t = torch.empty_like(t_vals) if need_t else None
    bh = _f32c(cfg.box_half)
    a = _render_args(rays_o, rays_d, t_vals, xyz_div, z_scale, latent, packed, cfg.frame, cfg.xyz_mul, cfg.z_mode, cfg.flags,
                     cfg.rays_per_obj, cfg.n_samples, cfg.shape_blocks, cfg.texture_blocks,
                     resolve_precision(cfg.precision, cfg.shape_blocks, cfg.texture_blocks, cfg.rays_per_obj * cfg.n_samples, backward=True),
                     box_half=bh, rng=cfg.rng)
    ws_bytes = _lib.lib().snr_render_bwd_ws_bytes(C.byref(a))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_render_bwd(C.byref(a), _p(sig), _p(rgbs), _p(masks), _p(d_rgb), _p(d_depth), _p(d_acc), _p(d_lat), _p(d_o), _p(d_d),
                                        _p(d_t), _p(ws), ws_bytes, _stream(dev)), "snr_render_bwd")
    return d_o, d_d, d_t, d_lat


# ------------------------------------------------------------------------------------ encode
def encode(rays_o, rays_d, t_vals, xyz_div, z_scale, cfg: RenderCfg, want_pe=False, want_hit=False):
    """Sample points of a ray packet: xyz (N,S,3), viewdir (N,S,3), z (N,S) [, PE(xyz) (N,S,63), PE(dir) (N,27)] [, hit (N) bool: the
    rays that meet their box, Z_BOX]."""
    rays_o, rays_d, t_vals, xyz_div, z_scale = [_f32c(t) for t in (rays_o, rays_d, t_vals, xyz_div, z_scale)]
    _need_gpu(rays_o, rays_d, t_vals, xyz_div, z_scale)
    dev = rays_o.device
    N, S = rays_o.shape[0], cfg.n_samples
    xyz = torch.empty(N, S, 3, device=dev)
    vd = torch.empty(N, S, 3, device=dev)
    z = torch.empty(N, S, device=dev)
    pe = torch.empty(N, S, 63, device=dev) if want_pe else None
    ped = torch.empty(N, 27, device=dev) if want_pe else None
    hit = torch.empty(N, dtype=torch.uint8, device=dev) if want_hit else None
    _box_rng(cfg, t_vals, dev, N * S)
    bh = _f32c(cfg.box_half)
    a = _render_args(rays_o, rays_d, t_vals, xyz_div, z_scale, None, None, cfg.frame, cfg.xyz_mul, cfg.z_mode, cfg.flags,
                     cfg.rays_per_obj, S, cfg.shape_blocks, cfg.texture_blocks, box_half=bh, rng=cfg.rng)
    with torch.cuda.device(dev):
        check(_lib.lib().snr_encode_fwd(C.byref(a), _p(xyz), _p(vd), _p(z), _p(pe), _p(ped), _p(hit), _stream(dev)), "snr_encode_fwd")
    out = (xyz, vd, z, pe, ped) if want_pe else (xyz, vd, z)
    return out + (hit.bool(),) if want_hit else out
