"""CPU oracle for the SUP-NeRF volumetric-rendering hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``sup-nerf_amd/``) never calls anything in here
and fails loudly when its HIP library is missing.

It is a from-scratch restatement, in plain PyTorch fp32 on the CPU, of what the
reference computes on the path

    rays -> stratified samples -> coordinate transforms -> positional encoding
         -> code-conditioned MLP decoder -> alpha composite

Every function cites the reference lines (relative to /root/reference) it
restates.  Differences from the reference are limited to *how* randomness is
supplied: the reference draws from the global torch / numpy / python RNGs
inside the functions; here every random quantity (``jitter``, ray permutation,
symmetry flip) is an explicit argument so a test can inject the same numbers
into the oracle and into the HIP path.  When an argument is left ``None`` the
oracle draws it from the same global generator the reference would use.

Parity status: PINNED.  ``tests/golden/gen_golden.py`` imports the reference
itself in the build container, runs it on seeded inputs, checks this file
against it and commits the input/output vectors under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks this file against those vectors
wherever the tests run (the reference itself never travels).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# decoder (src/model_supnerf.py:155-161,184-199,241-269 == src/model_codenerf.py)
# --------------------------------------------------------------------------

LAST_DELTA = 1e10      # src/utils.py:209  width of the last sample interval
TRANS_EPS = 1e-10      # src/utils.py:211  bias added to the per-sample transmittance


def positional_encoding(x: Tensor, degree: int) -> Tensor:
    """[x | sin(2^0 x) sin(2^1 x) .. | cos(2^0 x) ..], frequency-major then xyz.

    Restates ``PE`` (src/model_supnerf.py:155-161, src/model_codenerf.py:4-10).
    No factor of pi.  Output width 3 + 6*degree.
    """
    scaled = [x * (2.0 ** i) for i in range(degree)]
    arg = torch.cat(scaled, dim=-1)
    return torch.cat([x, torch.sin(arg), torch.cos(arg)], dim=-1)


def decoder_param_names(shape_blocks: int = 3, texture_blocks: int = 1) -> Sequence[str]:
    """State-dict keys of the decoder in the order the reference registers them
    (src/model_supnerf.py:184-199)."""
    names = ["encoding_xyz.0"]
    for j in range(1, shape_blocks + 1):
        names += [f"shape_latent_layer_{j}.0", f"shape_layer_{j}.0"]
    names += ["encoding_shape", "sigma.0", "encoding_viewdir.0"]
    for j in range(1, texture_blocks + 1):
        names += [f"texture_latent_layer_{j}.0", f"texture_layer_{j}.0"]
    names += ["rgb.0", "rgb.2"]
    out = []
    for n in names:
        out += [n + ".weight", n + ".bias"]
    return out


def init_decoder_params(shape_blocks: int = 3, texture_blocks: int = 1, W: int = 256,
                        latent_dim: int = 256, num_xyz_freq: int = 10, num_dir_freq: int = 4,
                        seed: int = 0, sigma_bias: Optional[float] = -2.0) -> Dict[str, Tensor]:
    """Seeded random decoder weights with ``nn.Linear``'s default init
    (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight
    and bias), in reference state-dict naming.  ``sigma_bias`` shifts the density
    head so alpha spans (0,1)."""
    g = torch.Generator().manual_seed(seed)
    d_xyz, d_dir = 3 + 6 * num_xyz_freq, 3 + 6 * num_dir_freq

    def lin(n_out, n_in):
        bound = 1.0 / math.sqrt(n_in)
        w = (torch.rand(n_out, n_in, generator=g) * 2 - 1) * bound
        b = (torch.rand(n_out, generator=g) * 2 - 1) * bound
        return w, b

    p: Dict[str, Tensor] = {}

    def put(name, n_out, n_in):
        p[name + ".weight"], p[name + ".bias"] = lin(n_out, n_in)

    put("encoding_xyz.0", W, d_xyz)
    for j in range(1, shape_blocks + 1):
        put(f"shape_latent_layer_{j}.0", W, latent_dim)
        put(f"shape_layer_{j}.0", W, W)
    put("encoding_shape", W, W)
    put("sigma.0", 1, W)
    put("encoding_viewdir.0", W, W + d_dir)
    for j in range(1, texture_blocks + 1):
        put(f"texture_latent_layer_{j}.0", W, latent_dim)
        put(f"texture_layer_{j}.0", W, W)
    put("rgb.0", W // 2, W)
    put("rgb.2", 3, W // 2)
    if sigma_bias is not None:
        p["sigma.0.bias"] = torch.full((1,), float(sigma_bias))
    return p


def _count_blocks(params: Dict[str, Tensor]) -> Tuple[int, int]:
    sb = sum(1 for k in params if k.startswith("shape_layer_") and k.endswith(".weight"))
    tb = sum(1 for k in params if k.startswith("texture_layer_") and k.endswith(".weight"))
    return sb, tb


def latent_terms(params: Dict[str, Tensor], shape_code: Tensor, texture_code: Tensor) -> Tensor:
    """Per-object latent vectors z_j = ReLU(Lin_j(code)) that the decoder adds to
    the hidden state before every shape / texture block
    (src/model_supnerf.py:253,261).  Returns (B, shape_blocks+texture_blocks, W)."""
    sb, tb = _count_blocks(params)
    outs = []
    for j in range(1, sb + 1):
        outs.append(F.relu(F.linear(shape_code, params[f"shape_latent_layer_{j}.0.weight"],
                                    params[f"shape_latent_layer_{j}.0.bias"])))
    for j in range(1, tb + 1):
        outs.append(F.relu(F.linear(texture_code, params[f"texture_latent_layer_{j}.0.weight"],
                                    params[f"texture_latent_layer_{j}.0.bias"])))
    return torch.stack(outs, dim=1)


class _ReluWithGivenMask(torch.autograd.Function):
    """relu(x) whose DERIVATIVE is a given 0/1 mask instead of [x > 0] (test infrastructure, see ``decoder_forward``)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return torch.relu(x)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


_GIVEN_RELU_MASKS = None


class given_relu_masks:
    """``with given_relu_masks(masks): ...`` -- every ``decoder_forward`` call inside (however deep: the render functions call it) takes
    ``masks`` as its ``relu_masks``.  Tests only."""

    def __init__(self, masks):
        self.masks = masks

    def __enter__(self):
        global _GIVEN_RELU_MASKS
        self.prev, _GIVEN_RELU_MASKS = _GIVEN_RELU_MASKS, self.masks

    def __exit__(self, *exc):
        global _GIVEN_RELU_MASKS
        _GIVEN_RELU_MASKS = self.prev


def decoder_forward(params: Dict[str, Tensor], xyz: Tensor, viewdir: Tensor,
                    shape_code: Tensor, texture_code: Tensor,
                    num_xyz_freq: int = 10, num_dir_freq: int = 4,
                    relu_masks: Optional[Sequence[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """sigma (N,S,1), rgb (N,S,3) for xyz/viewdir (N,S,3) and codes (B,latent).

    Restates ``SUPNeRF.forward`` (src/model_supnerf.py:241-269) ==
    ``CodeNeRF.forward`` (src/model_codenerf.py:39-63).  Rays are object-major:
    ray r belongs to object ``r // (N // B)`` (src/model_supnerf.py:246-249).

    ``relu_masks`` (tests only; None = the reference's computation): one 0/1 tensor per ReLU layer of the per-point chain in
    forward order (encoding_xyz, shape layers, encoding_viewdir, texture layers, rgb.0), each shaped like that layer's output.
    The forward values are unchanged; the backward of each ReLU uses the given mask as its derivative.  A hidden unit whose
    pre-activation sits within rounding of zero lands on either side of the ReLU depending on the summation order, which moves
    that point's gradient by percents in ANY two correct implementations; with the masks an implementation saved, the oracle
    differentiates the same piecewise-linear function as that implementation, and the comparison can be tight.
    """
    relu_i = [0]
    if relu_masks is None:
        relu_masks = _GIVEN_RELU_MASKS

    def relu(t):
        if relu_masks is None:
            return F.relu(t)
        m = relu_masks[relu_i[0]]
        relu_i[0] += 1
        return _ReluWithGivenMask.apply(t, m.reshape(t.shape))

    sb, tb = _count_blocks(params)
    n_ray = xyz.shape[0]
    n_obj = shape_code.shape[0]
    per_obj = int(n_ray / n_obj)
    # (B, L) -> (B*per_obj, 1, L): every ray of object b sees code b
    shape_rows = shape_code.repeat_interleave(per_obj, dim=0).unsqueeze(1)
    tex_rows = texture_code.repeat_interleave(per_obj, dim=0).unsqueeze(1)

    def lin(name, t):
        return F.linear(t, params[name + ".weight"], params[name + ".bias"])

    h = relu(lin("encoding_xyz.0", positional_encoding(xyz, num_xyz_freq)))
    for j in range(1, sb + 1):
        z = F.relu(lin(f"shape_latent_layer_{j}.0", shape_rows))
        h = relu(lin(f"shape_layer_{j}.0", h + z))
    h = lin("encoding_shape", h)                       # no activation
    sigma = F.softplus(lin("sigma.0", h))              # beta=1, threshold=20
    h = relu(lin("encoding_viewdir.0",
                 torch.cat([h, positional_encoding(viewdir, num_dir_freq)], dim=-1)))
    for j in range(1, tb + 1):
        z = F.relu(lin(f"texture_latent_layer_{j}.0", tex_rows))
        h = relu(lin(f"texture_layer_{j}.0", h + z))
    rgb = lin("rgb.2", relu(lin("rgb.0", h)))          # raw linear output, no sigmoid
    return sigma, rgb


# --------------------------------------------------------------------------
# ray generation (src/utils.py:107-151)
# --------------------------------------------------------------------------

def _rays_from_pixels(K: Tensor, c2w: Tensor, px: Tensor, py: Tensor) -> Tuple[Tensor, Tensor]:
    cx, cy, fx, fy = K[0, 2], K[1, 2], K[0, 0], K[1, 1]
    cam_dir = torch.stack([(px - cx) / fx, (py - cy) / fy, torch.ones_like(px)], dim=-1)
    cam_dir = cam_dir.type_as(c2w)
    # world_dir[c] = sum_k cam_dir[k] * R[c,k]   (src/utils.py:131)
    world = (cam_dir[..., None, :] * c2w[..., :3, :3]).sum(-1)
    unit = world / torch.norm(world, dim=-1, keepdim=True)
    origin = c2w[..., :3, -1].expand(world.shape)
    return origin.reshape(-1, 3), unit.reshape(-1, 3)


def pixel_rays(K: Tensor, c2w: Tensor, roi, uv_steps: Optional[Sequence[int]] = None) -> Tuple[Tensor, Tensor]:
    """Rays through a regular pixel grid over roi=[xmin,ymin,xmax,ymax]; rays are
    ordered row-major over (y, x).  Restates ``get_rays`` (src/utils.py:107-135)."""
    x0, y0, x1, y1 = [int(v) for v in roi]
    nx, ny = (uv_steps[0], uv_steps[1]) if uv_steps is not None else (x1 - x0, y1 - y0)
    xs = torch.linspace(x0, x1 - 1, nx)
    ys = torch.linspace(y0, y1 - 1, ny)
    px = xs[None, :].expand(ny, nx)
    py = ys[:, None].expand(ny, nx)
    return _rays_from_pixels(K, c2w, px, py)


def pixel_rays_at(K: Tensor, c2w: Tensor, x_vec: np.ndarray, y_vec: np.ndarray) -> Tuple[Tensor, Tensor]:
    """Rays through listed pixel coordinates.  Restates ``get_rays_specified``
    (src/utils.py:138-151)."""
    return _rays_from_pixels(K, c2w, torch.from_numpy(np.asarray(x_vec)), torch.from_numpy(np.asarray(y_vec)))


# --------------------------------------------------------------------------
# sampling
# --------------------------------------------------------------------------

def sphere_bounds(cam_pose: Tensor, obj_diag: float) -> Tuple[float, float]:
    """near/far = |camera centre| -/+ diag/2, detached python floats
    (src/utils.py:468-469)."""
    dist = np.linalg.norm(cam_pose[:, -1].tolist())
    return dist - obj_diag / 2, dist + obj_diag / 2


def shared_depth_samples(near: float, far: float, n_samples: int, jitter: Optional[Tensor] = None) -> Tensor:
    """Family A: one (S,) vector of stratified depths shared by all rays.
    Restates the z part of ``sample_from_rays`` (src/utils.py:162-164).
    ``jitter`` is the ``torch.rand(S)`` draw."""
    half = (far - near) / (2 * n_samples)
    z = torch.linspace(near + half, far - half, n_samples)
    if jitter is None:
        jitter = torch.rand(n_samples)
    return z + jitter * (far - near) / (2 * n_samples)


def points_on_rays(ro: Tensor, vd: Tensor, z: Tensor) -> Tuple[Tensor, Tensor]:
    """xyz (N,S,3) = o + z d and the direction repeated along S
    (src/utils.py:165-166).  ``z`` is (S,) or (N,S)."""
    zz = z.type_as(ro)
    if zz.dim() == 1:
        xyz = ro[:, None, :] + vd[:, None, :] * zz[None, :, None]
    else:
        xyz = ro[:, None, :] + vd[:, None, :] * zz[:, :, None]
    return xyz, vd[:, None, :].repeat(1, zz.shape[-1], 1)


def unit_interval_samples(near: Tensor, far: Tensor, n_samples: int, jitter: Optional[Tensor] = None) -> Tensor:
    """Family B: per-ray stratified depths between per-ray near/far (N,1).
    Restates ``NeRFRenderer.sample_from_ray`` (src/renderer.py:27-41) ==
    ``sample_from_rays_v2`` (src/utils.py:170-184).  ``jitter`` is the
    ``rand_like`` draw of shape (N,S)."""
    step = 1.0 / n_samples
    n = near.shape[0]
    t = torch.linspace(0, 1 - step, n_samples, device=near.device)[None, :].repeat(n, 1)
    if jitter is None:
        jitter = torch.rand_like(t)
    t = t + jitter * step
    return near * (1 - t) + far * t


def slab_intersect(o: Tensor, d: Tensor, bmin: Tensor, bmax: Tensor):
    """Ray / axis-aligned-box slab test.  Returns (t_near, t_far, hit) for ALL
    rays (the reference returns the hit subset; callers scatter it back).
    Restates ``ray_box_intersection_tensor`` (src/utils.py:283-327); min/max are
    NaN-propagating like torch.minimum/maximum."""
    inv = torch.reciprocal(d)
    ta = (bmin - o) * inv
    tb = (bmax - o) * inv
    lo = torch.minimum(ta, tb)
    hi = torch.maximum(ta, tb)
    t_near = torch.maximum(torch.maximum(lo[..., 0], lo[..., 1]), lo[..., 2])
    t_far = torch.minimum(torch.minimum(hi[..., 0], hi[..., 1]), hi[..., 2])
    hit = t_far > t_near
    hit = torch.logical_and(hit, (t_far * hit) > 0)
    return t_near, t_far, hit


def aabb_sampled_rays(rays_o: Tensor, viewdir: Tensor, obj_sz, n_samples: int,
                      jitter: Optional[Tensor] = None, detach_bounds: bool = False):
    """Family B sample preparation.  Restates ``NeRFRenderer.prepare_sampled_rays``
    (src/renderer.py:91-115).  obj_sz = (w, l, h); box half extents are
    (l, w, h)/diag in the frame where origins are divided by diag/2; rays that
    miss get near=far=-1; z_vals is the metric distance |xyz-o|*diag/2.
    ``detach_bounds`` reproduces ``render_rays_v3`` where the slab test runs in
    numpy float64-free detached form (src/renderer.py:425-432)."""
    obj_sz = np.asarray(obj_sz)
    diag = np.linalg.norm(obj_sz).astype(np.float32)
    w, l, h = [float(v) for v in obj_sz]
    half = torch.tensor([l / diag, w / diag, h / diag], dtype=torch.float32, device=rays_o.device)
    o_n = rays_o / (diag / 2)
    src_o, src_d = (o_n.detach(), viewdir.detach()) if detach_bounds else (o_n, viewdir)
    t_near, t_far, hit = slab_intersect(src_o, src_d, -half.expand_as(src_o), half.expand_as(src_o))
    minus1 = torch.full_like(t_near, -1.0)
    near = torch.where(hit, t_near, minus1)[:, None]
    far = torch.where(hit, t_far, minus1)[:, None]
    z_unit = unit_interval_samples(near, far, n_samples, jitter)
    xyz = o_n[:, None, :] + z_unit[:, :, None] * viewdir[:, None, :]
    vd = viewdir[:, None, :].repeat(1, n_samples, 1)
    z_vals = torch.norm((xyz - o_n[:, None, :]) * (diag / 2), p=2, dim=-1)
    return xyz, vd, z_vals, hit


# --------------------------------------------------------------------------
# coordinate transforms (src/utils.py:472-495)
# --------------------------------------------------------------------------

def object_frame_transforms(xyz: Tensor, viewdir: Tensor, sym_flip: bool = False,
                            kitti2nusc: bool = False, shapenet_obj_cood: bool = False):
    """Optional, in the reference's order: mirror y (sym_aug draw came up > 0.5),
    KITTI->nuScenes axis rotation R_x, nuScenes->ShapeNet (x,y,z)->(-y,x,z)."""
    if sym_flip:                                   # src/utils.py:475-477
        m = torch.tensor([1.0, -1.0, 1.0], dtype=xyz.dtype, device=xyz.device)
        xyz, viewdir = xyz * m, viewdir * m
    if kitti2nusc:                                 # src/utils.py:480-488: (x,y,z)->(x,z,-y)
        xyz = torch.stack([xyz[..., 0], xyz[..., 2], -xyz[..., 1]], dim=-1)
        viewdir = torch.stack([viewdir[..., 0], viewdir[..., 2], -viewdir[..., 1]], dim=-1)
    if shapenet_obj_cood:                          # src/utils.py:491-495
        xyz = torch.stack([-xyz[..., 1], xyz[..., 0], xyz[..., 2]], dim=-1)
        viewdir = torch.stack([-viewdir[..., 1], viewdir[..., 0], viewdir[..., 2]], dim=-1)
    return xyz, viewdir


# --------------------------------------------------------------------------
# alpha composite (src/utils.py:202-233, src/renderer.py:43-65,355-379)
# --------------------------------------------------------------------------

def composite(sigmas: Tensor, rgbs: Tensor, z_vals: Tensor, white_bkgd: bool = False):
    """Density -> transmittance alpha composite along the last sample axis.

    sigmas (..., S) or (..., S, 1); rgbs (..., S, 3); z_vals broadcastable to
    (..., S): shared (S,), per object (B,1,S) or per ray (N,S).  Returns
    rgb (...,3), depth (...), acc_trans (...) where acc_trans is the product of
    the first S-1 per-sample transmittances (it excludes the last, 1e10-wide
    sample).  One restatement covers ``volume_rendering2`` (src/utils.py:202-217),
    ``volume_rendering_batch`` (:220-233), ``NeRFRenderer.volume_render``
    (src/renderer.py:43-65) and ``volume_rendering3`` (:355-379).
    """
    if sigmas.dim() == rgbs.dim():
        sigmas = sigmas.squeeze(-1)
    z = z_vals
    delta = z[..., 1:] - z[..., :-1]
    delta = torch.cat([delta, torch.full_like(delta[..., :1], LAST_DELTA)], dim=-1)
    alpha = 1 - torch.exp(-torch.relu(sigmas) * delta)
    trans = 1 - alpha + TRANS_EPS
    shifted = torch.cat([torch.ones_like(trans[..., :1]), trans], dim=-1)
    acc = torch.cumprod(shifted, dim=-1)[..., :-1]          # exclusive product
    w = alpha * acc
    rgb = (w[..., None] * rgbs).sum(-2)
    depth = (w * z).sum(-1)
    if white_bkgd:
        rgb = rgb + 1 - w.sum(-1)[..., None]
    return rgb, depth, acc[..., -1]


def volume_rendering2(sigmas, rgbs, z_vals):
    """src/utils.py:202-217 -- z (S,) shared, black background."""
    return composite(sigmas, rgbs, z_vals, white_bkgd=False)


def volume_rendering_batch(sigmas, rgbs, z_vals):
    """src/utils.py:220-233 -- sigmas (B,n,S,1), rgbs (B,n,S,3), z (B,S)."""
    return composite(sigmas, rgbs, z_vals[:, None, :], white_bkgd=False)


def volume_rendering3(sigmas, rgbs, z_vals, white_bkgd=False):
    """src/renderer.py:355-379 -- z (N,S) per ray."""
    return composite(sigmas, rgbs, z_vals, white_bkgd=white_bkgd)


# --------------------------------------------------------------------------
# target preparation (src/utils.py:447-456)
# --------------------------------------------------------------------------

def resize_targets(img: Tensor, mask_occ: Tensor, im_sz: int) -> Tuple[Tensor, Tensor]:
    """Bilinear resize (no antialias, torchvision 0.13 tensor semantics) of the
    target crop (h,w,3) and occupancy mask (h,w,1) to im_sz^2; the mask is
    truncated toward zero through int32 (src/utils.py:452)."""
    im = F.interpolate(img.permute(2, 0, 1)[None], size=(im_sz, im_sz), mode="bilinear",
                       align_corners=False)[0].permute(1, 2, 0)
    mk = F.interpolate(mask_occ.permute(2, 0, 1)[None], size=(im_sz, im_sz), mode="bilinear",
                       align_corners=False)[0].permute(1, 2, 0)
    mk = mk.type(torch.int32).type(torch.float32)
    return im, mk


# --------------------------------------------------------------------------
# drivers
# --------------------------------------------------------------------------

def _decode(params_or_model, xyz, viewdir, shapecode, texturecode):
    if callable(params_or_model):
        return params_or_model(xyz, viewdir, shapecode, texturecode)
    return decoder_forward(params_or_model, xyz, viewdir, shapecode, texturecode)


def render_rays_v2(model, img, mask_occ, cam_pose, obj_diag, K, roi, n_samples, shapecode, texturecode,
                   shapenet_obj_cood, sym_flip=False, kitti2nusc=False, im_sz=64, ray_ids=None, jitter=None):
    """Family A end to end.  Restates ``render_rays_v2`` (src/utils.py:435-502).
    ``model`` is a params dict or a callable decoder.  ``ray_ids`` replaces the
    np.random.permutation subset, ``jitter`` the torch.rand(S) draw, ``sym_flip``
    the outcome of the sym_aug coin."""
    rays_o, viewdir = pixel_rays(K, cam_pose, roi, uv_steps=[im_sz, im_sz])
    im, mk = resize_targets(img, mask_occ, im_sz)
    rgb_tgt, occ = im.reshape(-1, 3), mk.reshape(-1, 1)
    if ray_ids is not None:
        rays_o, viewdir, rgb_tgt, occ = rays_o[ray_ids], viewdir[ray_ids], rgb_tgt[ray_ids], occ[ray_ids]
    near, far = sphere_bounds(cam_pose, obj_diag)
    z = shared_depth_samples(near, far, n_samples, jitter).type_as(rays_o)
    xyz, vd = points_on_rays(rays_o, viewdir, z)
    xyz = xyz / obj_diag
    xyz, vd = object_frame_transforms(xyz, vd, sym_flip, kitti2nusc, shapenet_obj_cood)
    sig, rgb = _decode(model, xyz, vd, shapecode, texturecode)
    rgb_r, depth_r, acc_r = volume_rendering2(sig, rgb, z)
    return rgb_r, depth_r, acc_r, rgb_tgt, occ


def render_rays_specified(model, img, mask_occ, cam_pose, obj_diag, K, roi, x_vec, y_vec, n_samples,
                          shapecode, texturecode, shapenet_obj_cood, sym_flip=False, kitti2nusc=False, jitter=None):
    """Family A at listed pixels.  Restates ``render_rays_specified``
    (src/utils.py:504-551)."""
    rays_o, viewdir = pixel_rays_at(K, cam_pose, x_vec + int(roi[0]), y_vec + int(roi[1]))
    rgb_tgt = img[y_vec, x_vec, :]
    occ = mask_occ[y_vec, x_vec, :]
    near, far = sphere_bounds(cam_pose, obj_diag)
    z = shared_depth_samples(near, far, n_samples, jitter).type_as(rays_o)
    xyz, vd = points_on_rays(rays_o, viewdir, z)
    xyz = xyz / obj_diag
    xyz, vd = object_frame_transforms(xyz, vd, sym_flip, kitti2nusc, shapenet_obj_cood)
    sig, rgb = _decode(model, xyz, vd, shapecode, texturecode)
    rgb_r, depth_r, acc_r = volume_rendering2(sig, rgb, z)
    return rgb_r, depth_r, acc_r, rgb_tgt, occ


def prepare_pixel_samples(img, mask_occ, cam_pose, obj_diag, K, roi, n_rays, n_samples, shapenet_obj_cood,
                          sym_flip=False, im_sz=None, ray_ids=None, jitter=None):
    """Family A sample preparation used by the datasets/trainer.  Restates
    ``prepare_pixel_samples`` (src/utils.py:330-377)."""
    near, far = sphere_bounds(cam_pose, obj_diag)
    if im_sz is None:
        rays_o, viewdir = pixel_rays(K, cam_pose, roi)
    else:
        rays_o, viewdir = pixel_rays(K, cam_pose, roi, uv_steps=[im_sz, im_sz])
        img, mask_occ = resize_targets(img, mask_occ, im_sz)
    n_rays = int(np.minimum(rays_o.shape[0], n_rays))
    if ray_ids is None:
        ray_ids = np.random.permutation(rays_o.shape[0])[:n_rays]
    rays_o, viewdir = rays_o[ray_ids], viewdir[ray_ids]
    rgb_tgt = img.reshape(-1, 3)[ray_ids]
    occ = mask_occ.reshape(-1, 1)[ray_ids]
    z = shared_depth_samples(near, far, n_samples, jitter).type_as(rays_o)
    xyz, vd = points_on_rays(rays_o, viewdir, z)
    xyz = xyz / obj_diag
    xyz, vd = object_frame_transforms(xyz, vd, sym_flip, False, shapenet_obj_cood)
    return xyz, vd, z, rgb_tgt, occ


def render_full_img(model, cam_pose, obj_sz, K, roi, n_samples, shapecode, texturecode, shapenet_obj_cood,
                    out_depth=False, kitti2nusc=False, jitter=None):
    """Family A, every pixel of the roi.  Restates ``render_full_img``
    (src/utils.py:554-616), including its slab-wise decoder calls."""
    obj_diag = np.linalg.norm(obj_sz).astype(np.float32)
    rays_o, viewdir = pixel_rays(K, cam_pose, roi)
    near, far = sphere_bounds(cam_pose, obj_diag)
    z = shared_depth_samples(near, far, n_samples, jitter).type_as(rays_o)
    xyz, vd = points_on_rays(rays_o, viewdir, z)
    xyz = xyz / obj_diag
    xyz, vd = object_frame_transforms(xyz, vd, False, kitti2nusc, shapenet_obj_cood)
    h, w = int(roi[3] - roi[1]), int(roi[2] - roi[0])
    step = max(h, w)                       # slabs of max(roi_w, roi_h) rays (src/utils.py:591-597)
    parts = []
    for i in range(0, xyz.shape[0], step):
        sig, rgb = _decode(model, xyz[i:i + step], vd[i:i + step], shapecode, texturecode)
        parts.append(volume_rendering2(sig, rgb, z))
    rgb_r = torch.cat([p[0] for p in parts])
    depth_r = torch.cat([p[1] for p in parts])
    if out_depth:
        return rgb_r.reshape(h, w, 3), depth_r.reshape(h, w)
    return rgb_r.reshape(h, w, 3)


def nerf_renderer_render_rays(model, img, mask_occ, cam_pose, obj_sz, K, roi, shapecode, texturecode,
                              n_samples=64, white_bkgd=True, kitti2nusc=False, im_sz=64, ray_ids=None, jitter=None):
    """Family B end to end.  Restates ``NeRFRenderer.render_rays``
    (src/renderer.py:117-167)."""
    rays_o, viewdir = pixel_rays(K, cam_pose, roi, uv_steps=[im_sz, im_sz])
    im, mk = resize_targets(img, mask_occ, im_sz)
    rgb_tgt, occ = im.reshape(-1, 3), mk.reshape(-1, 1)
    if ray_ids is not None:
        rays_o, viewdir, rgb_tgt, occ = rays_o[ray_ids], viewdir[ray_ids], rgb_tgt[ray_ids], occ[ray_ids]
    xyz, vd, z_vals, _hit = aabb_sampled_rays(rays_o, viewdir, obj_sz, n_samples, jitter)
    xyz, vd = object_frame_transforms(xyz, vd, False, kitti2nusc, False)
    sig, rgb = _decode(model, xyz, vd, shapecode, texturecode)
    rgb_r, depth_r, acc_r = composite(sig, rgb, z_vals, white_bkgd=white_bkgd)
    return rgb_r, depth_r, acc_r, rgb_tgt, occ


def nerf_renderer_render_rays_specified(model, img, mask_occ, cam_pose, obj_sz, K, roi, x_vec, y_vec, shapecode, texturecode,
                                        n_samples=64, white_bkgd=True, kitti2nusc=False, jitter=None):
    """Family B at listed crop pixels.  Restates ``NeRFRenderer.render_rays_specified`` (src/renderer.py:169-201)."""
    rays_o, viewdir = pixel_rays_at(K, cam_pose, np.asarray(x_vec) + int(roi[0]), np.asarray(y_vec) + int(roi[1]))
    rgb_tgt, occ = img[y_vec, x_vec, :], mask_occ[y_vec, x_vec, :]
    xyz, vd, z_vals, _hit = aabb_sampled_rays(rays_o, viewdir, obj_sz, n_samples, jitter)
    xyz, vd = object_frame_transforms(xyz, vd, False, kitti2nusc, False)
    sig, rgb = _decode(model, xyz, vd, shapecode, texturecode)
    rgb_r, depth_r, acc_r = composite(sig, rgb, z_vals, white_bkgd=white_bkgd)
    return rgb_r, depth_r, acc_r, rgb_tgt, occ


def nerf_renderer_prepare_pixel_samples(img, mask_occ, cam_pose, obj_sz, K, roi, n_rays, n_samples=64, im_sz=None, ray_ids=None,
                                        jitter=None):
    """Pre-sampled family-B ray batch.  Restates ``NeRFRenderer.prepare_pixel_samples`` (src/renderer.py:203-236);
    ``ray_ids`` = the first n_rays entries of its ``np.random.permutation`` draw."""
    if im_sz is None:
        rays_o, viewdir = pixel_rays(K, cam_pose, roi)
    else:
        rays_o, viewdir = pixel_rays(K, cam_pose, roi, uv_steps=[im_sz, im_sz])
        img, mask_occ = resize_targets(img, mask_occ, im_sz)
    rays_o, viewdir = rays_o[ray_ids], viewdir[ray_ids]
    rgb_tgt, occ = img.reshape(-1, 3)[ray_ids], mask_occ.reshape(-1, 1)[ray_ids]
    xyz, vd, z_vals, _hit = aabb_sampled_rays(rays_o, viewdir, obj_sz, n_samples, jitter)
    return xyz, vd, z_vals, rgb_tgt, occ


def nerf_renderer_render_full_img(model, cam_pose, obj_sz, K, roi, shapecode, texturecode, n_samples=64, white_bkgd=True,
                                  out_depth=False, kitti2nusc=False, jitter=None):
    """Family B, every pixel of the roi, decoder called in slabs of max(roi_w, roi_h) rays.
    Restates ``NeRFRenderer.render_full_img`` (src/renderer.py:238-294)."""
    rays_o, viewdir = pixel_rays(K, cam_pose, roi)
    xyz, vd, z_vals, _hit = aabb_sampled_rays(rays_o, viewdir, obj_sz, n_samples, jitter)
    xyz, vd = object_frame_transforms(xyz, vd, False, kitti2nusc, False)
    h, w = int(roi[3] - roi[1]), int(roi[2] - roi[0])
    step = max(h, w)
    parts = []
    for i in range(0, xyz.shape[0], step):
        sig, rgb = _decode(model, xyz[i:i + step], vd[i:i + step], shapecode, texturecode)
        parts.append(composite(sig, rgb, z_vals[i:i + step], white_bkgd=white_bkgd))
    rgb_r = torch.cat([p_[0] for p_ in parts]).reshape(h, w, 3)
    if out_depth:
        return rgb_r, torch.cat([p_[1] for p_ in parts]).reshape(h, w)
    return rgb_r


def turntable_poses(radius=40.0, tilt=np.pi / 6, pan_num=8):
    """The camera-in-object poses (3,4) of ``render_virtual_imgs`` (src/utils.py:633-650 == src/renderer.py:309-326)."""
    cam_init = np.asarray([[0, 0, 1, -radius], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 1]]).astype(np.float32)
    cam_tilt = np.asarray([[np.cos(tilt), 0, np.sin(tilt), 0], [0, 1, 0, 0], [-np.sin(tilt), 0, np.cos(tilt), 0],
                           [0, 0, 0, 1]]).astype(np.float32) @ cam_init
    out = []
    for pan in np.linspace(0, 2 * np.pi, pan_num, endpoint=False):
        pose = np.asarray([[np.cos(pan), -np.sin(pan), 0, 0], [np.sin(pan), np.cos(pan), 0, 0], [0, 0, 1, 0],
                           [0, 0, 0, 1]]).astype(np.float32) @ cam_tilt
        out.append(torch.from_numpy(pose[:3, :]))
    return out


def virtual_roi(K, img_sz):
    x0, y0 = K[0, 2] - img_sz / 2, K[1, 2] - img_sz / 2
    return np.asarray([x0, y0, x0 + img_sz, y0 + img_sz]).astype(np.int64)


def render_virtual_imgs(model, obj_sz, K, n_samples, shapecode, texturecode, shapenet_obj_cood, radius=40., tilt=np.pi / 6,
                        pan_num=8, img_sz=128, kitti2nusc=False, jitters=None):
    """Family A turntable views WITHOUT the axis arrows the reference draws with cv2 (src/utils.py:619-672)."""
    roi = virtual_roi(K, img_sz)
    return [render_full_img(model, pose, obj_sz, K, roi, n_samples, shapecode, texturecode, shapenet_obj_cood, kitti2nusc=kitti2nusc,
                            jitter=None if jitters is None else jitters[i]) for i, pose in enumerate(turntable_poses(radius, tilt, pan_num))]


def nerf_renderer_render_virtual_imgs(model, obj_sz, K, shapecode, texturecode, n_samples=64, white_bkgd=True, radius=40.,
                                      tilt=np.pi / 6, pan_num=8, img_sz=128, kitti2nusc=False, jitters=None):
    """Family B turntable views without the cv2 arrows (src/renderer.py:296-352)."""
    roi = virtual_roi(K, img_sz)
    return [nerf_renderer_render_full_img(model, pose, obj_sz, K, roi, shapecode, texturecode, n_samples, white_bkgd,
                                          kitti2nusc=kitti2nusc, jitter=None if jitters is None else jitters[i])
            for i, pose in enumerate(turntable_poses(radius, tilt, pan_num))]


def srn_rays(H, W, focal, c2w):
    """``get_rays_srn`` (src/utils.py:94-104): ShapeNet-SRN camera convention (y up, looking down -z)."""
    xs, ys = torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H)
    px, py = xs[None, :].expand(H, W), ys[:, None].expand(H, W)
    cam = torch.stack([(px - W * .5) / focal, -(py - H * .5) / focal, -torch.ones_like(px)], -1)
    world = (cam[..., None, :] * c2w[:3, :3]).sum(-1)
    unit = world / torch.norm(world, dim=-1, keepdim=True)
    return c2w[:3, -1].expand(world.shape).reshape(-1, 3), unit.reshape(-1, 3)


def volume_rendering_legacy(sigmas, rgbs, z_vals):
    """``volume_rendering`` (src/utils.py:187-199): two outputs and NO relu on the densities."""
    deltas = torch.cat([z_vals[1:] - z_vals[:-1], torch.ones_like(z_vals[:1]) * LAST_DELTA])
    alphas = 1 - torch.exp(-sigmas.squeeze(-1) * deltas)
    trans = 1 - alphas + TRANS_EPS
    acc = torch.cumprod(torch.cat([torch.ones_like(trans[..., :1]), trans], -1), -1)[..., :-1]
    w = alphas * acc
    return torch.sum(w.unsqueeze(-1) * rgbs, -2), torch.sum(w * z_vals, -1)


def render_rays_v3(model, img, mask_occ, cam_pose, obj_wlh, K, roi, n_samples, shapecode, texturecode,
                   shapenet_obj_cood, sym_flip=False, kitti2nusc=False, im_sz=64, ray_ids=None,
                   adjust_scale=1.0, jitter=None):
    """Family B function form with detached (numpy) box bounds, scale adjust,
    black background.  Restates ``render_rays_v3`` (src/renderer.py:382-473)."""
    # The reference builds a default NeRFRenderer() (64 samples) for the depths but
    # repeats viewdir n_samples times (src/renderer.py:393,434,437): any other
    # n_samples raises inside the decoder's cat, so 64 is the only valid value.
    if n_samples != 64:
        raise ValueError("render_rays_v3 only works with n_samples == 64 (reference behaviour)")
    rays_o, viewdir = pixel_rays(K, cam_pose, roi, uv_steps=[im_sz, im_sz])
    im, mk = resize_targets(img, mask_occ, im_sz)
    rgb_tgt, occ = im.reshape(-1, 3), mk.reshape(-1, 1)
    if ray_ids is not None:
        rays_o, viewdir, rgb_tgt, occ = rays_o[ray_ids], viewdir[ray_ids], rgb_tgt[ray_ids], occ[ray_ids]
    xyz, vd, z_vals, _hit = aabb_sampled_rays(rays_o, viewdir, obj_wlh, n_samples, jitter, detach_bounds=True)
    xyz = xyz * adjust_scale
    xyz, vd = object_frame_transforms(xyz, vd, sym_flip, kitti2nusc, shapenet_obj_cood)
    sig, rgb = _decode(model, xyz, vd, shapecode, texturecode)
    rgb_r, depth_r, acc_r = composite(sig, rgb, z_vals, white_bkgd=False)
    return rgb_r, depth_r, acc_r, rgb_tgt, occ


# --------------------------------------------------------------------------
# loss / metric tail (src/optimizer_nuscenes.py:729-744)
# --------------------------------------------------------------------------

def optimise_losses(rgb_rays, acc_trans_rays, rgb_tgt, occ_pixels, loss_occ_coef=0.1):
    """loss, loss_rgb, loss_occ and the foreground-only PSNR the reference logs."""
    a = torch.abs(occ_pixels)
    denom = a.sum() + 1e-9
    loss_rgb = (((rgb_rays - rgb_tgt) ** 2) * a).sum() / denom
    loss_occ = (torch.exp(-occ_pixels * (0.5 - acc_trans_rays.unsqueeze(-1))) * a).sum() / denom
    loss = loss_rgb + loss_occ_coef * loss_occ
    fg = occ_pixels.clone()
    fg[occ_pixels < 0] = 0
    mse_fg = (((rgb_rays - rgb_tgt) ** 2) * fg).sum() / (fg.sum() + 1e-9)
    psnr = -10.0 * torch.log10(mse_fg)
    return loss, loss_rgb, loss_occ, psnr


def training_losses(params, xyz_batch, viewdir_batch, shapecode_batch, texturecode_batch, z_vals_batch, rgb_tgt_batch,
                    occ_pixels_batch, loss_occ_coef=0.1):
    """NeRF half of ParallelModel.forward (src/trainer_unified_nuscenes.py:117-148): per-object masked losses, then the
    mean over the objects of the batch.  Returns (loss_total, loss_rgb, loss_occ, loss_reg, psnr)."""
    B, n, S = xyz_batch.shape[:3]
    sig, rgb = decoder_forward(params, xyz_batch.flatten(0, 1), viewdir_batch.flatten(0, 1), shapecode_batch, texturecode_batch)
    rgb_rays, _, acc = volume_rendering_batch(sig.view(B, n, S, 1), rgb.view(B, n, S, 3), z_vals_batch)
    a = torch.abs(occ_pixels_batch)
    denom = torch.sum(a, dim=[-2, -1]) + 1e-9
    loss_rgb = torch.sum((rgb_rays - rgb_tgt_batch) ** 2 * a, dim=[-2, -1]) / denom
    loss_occ = torch.sum(torch.exp(-occ_pixels_batch * (0.5 - acc.unsqueeze(-1))) * a, dim=[-2, -1]) / denom
    loss_reg = torch.norm(shapecode_batch, dim=-1) + torch.norm(texturecode_batch, dim=-1)
    psnr = -10.0 * torch.log(loss_rgb.mean()) / math.log(10.0)
    total = loss_rgb.mean() + loss_occ_coef * loss_occ.mean()
    return total, loss_rgb.mean(), loss_occ.mean(), loss_reg.mean(), psnr.detach()


# --------------------------------------------------------------------------
# multi-object scene compositing (scripts/demo.py:425-579, OptimizerDemo.vis_scene)
# --------------------------------------------------------------------------

def box_corners(obj_poses: Tensor, wlh: Tensor) -> Tensor:
    """(Nb,3,8) corners of oriented boxes, nuScenes order (x forward, y left, z up).
    Restates ``corners_of_box_batch(..., is_kitti=False)`` (src/utils.py:1110-1148)."""
    sx = torch.tensor([1, 1, 1, 1, -1, -1, -1, -1], dtype=wlh.dtype)
    sy = torch.tensor([1, -1, -1, 1, 1, -1, -1, 1], dtype=wlh.dtype)
    sz = torch.tensor([1, 1, -1, -1, 1, 1, -1, -1], dtype=wlh.dtype)
    w, l, h = wlh[:, 0:1], wlh[:, 1:2], wlh[:, 2:3]
    local = torch.stack([l / 2 * sx, w / 2 * sy, h / 2 * sz], dim=1)            # (Nb,3,8)
    return torch.matmul(obj_poses[:, :, :3], local) + obj_poses[:, :, 3:4]


def project_points(points: Tensor, K: Tensor) -> Tensor:
    """Perspective projection of (Nb,3,n) camera-frame points, (Nb,3,n) with rows (u, v, 1).
    Restates ``view_points_batch(points, K, normalize=True)`` (src/utils.py:1032-1075)."""
    uvw = torch.matmul(K, points)
    return uvw / uvw[:, 2:3, :]


def clip_roi(roi: Tensor, H: int, W: int) -> Tensor:
    """``roi_process(roi, H, W, roi_margin=0, sq_pad=False)`` (src/utils.py:1392-1415): clip to the image."""
    out = roi.clone()
    out[0:2] = torch.maximum(out[0:2], torch.as_tensor(0))
    out[2] = torch.minimum(out[2], torch.as_tensor(W - 1))
    out[3] = torch.minimum(out[3], torch.as_tensor(H - 1))
    return out


def scene_rays(obj_poses: Tensor, obj_wlh: Tensor, K: Tensor, H: int, W: int, manipulation=(0.0, 0.0, 0.0),
               rend_aabb: bool = True):
    """Per-pixel, per-object ray table (H,W,Nb,8) = [origin/(diag/2) (3), unit dir (3), near, far] with -1 where an
    object does not cover the pixel, plus the pixel mask of rays that hit anything and the diagonals.
    Restates scripts/demo.py:437-523."""
    Nb = obj_poses.shape[0]
    all_rays = torch.ones((H, W, Nb, 8), dtype=torch.float32) * (-1)
    poses = obj_poses.clone()
    poses[:, :, 3] += torch.tensor(manipulation, dtype=torch.float32).unsqueeze(0)
    uv = project_points(box_corners(poses, obj_wlh), K.unsqueeze(0).repeat(Nb, 1, 1))
    rois = torch.stack([uv[:, 0].min(dim=1)[0], uv[:, 1].min(dim=1)[0], uv[:, 0].max(dim=1)[0], uv[:, 1].max(dim=1)[0]], dim=1)
    rois = rois.type(torch.int32)
    diags = []
    for i in range(Nb):
        roi = clip_roi(rois[i], H, W)
        R_c2o = poses[i, :3, :3].transpose(0, 1)
        cam_pose = torch.cat([R_c2o, -R_c2o @ poses[i, :3, 3:4]], dim=1)
        rays_o, viewdir = pixel_rays(K, cam_pose, roi)
        wlh = np.asarray(obj_wlh[i])
        diag = np.linalg.norm(wlh).astype(np.float32)
        diags.append(diag)
        x0, y0, x1, y1 = [int(v) for v in roi]
        all_rays[y0:y1, x0:x1, i, :3] = rays_o.view(y1 - y0, x1 - x0, -1) / (diag / 2)
        all_rays[y0:y1, x0:x1, i, 3:6] = viewdir.view(y1 - y0, x1 - x0, -1)
        if rend_aabb:
            ow, ol, oh = wlh
            bmax = np.asarray([ol / diag, ow / diag, oh / diag]).reshape(1, 3).repeat(rays_o.shape[0], axis=0)
            o_np = rays_o.numpy() / (diag / 2)                 # numpy slab test, float64 box like the reference's np.asarray
            t_near, t_far, hit = slab_intersect(torch.from_numpy(o_np), torch.from_numpy(viewdir.numpy().astype(o_np.dtype)),
                                                torch.from_numpy(-bmax), torch.from_numpy(bmax))
            near = all_rays[y0:y1, x0:x1, i, 6].flatten(0, 1)
            far = all_rays[y0:y1, x0:x1, i, 7].flatten(0, 1)
            near[hit] = t_near[hit].type(torch.float32)
            far[hit] = t_far[hit].type(torch.float32)
            all_rays[y0:y1, x0:x1, i, 6] = near.view(y1 - y0, x1 - x0)
            all_rays[y0:y1, x0:x1, i, 7] = far.view(y1 - y0, x1 - x0)
        else:
            dist = torch.linalg.norm(cam_pose[:, -1])
            all_rays[y0:y1, x0:x1, i, 6] = (dist - diag / 2) / (diag / 2)
            all_rays[y0:y1, x0:x1, i, 7] = (dist + diag / 2) / (diag / 2)
    diags = torch.tensor(diags, dtype=torch.float32)
    valid = (all_rays[:, :, :, 7].view(H * W, Nb) - all_rays[:, :, :, 6].view(H * W, Nb)).max(-1)[0] > 0
    return all_rays, valid, diags


def scene_composite(sigmas: Tensor, rgbs: Tensor, z_vals: Tensor, white_bkgd: bool = True):
    """Merge the Nb*S samples of a pixel by depth, then ``volume_rendering3``.  sigmas, z_vals (P, n); rgbs (P, n, 3).
    Restates scripts/demo.py:555-565 including its scatter through ``searchsorted`` (equal depths collapse onto one
    slot and leave zero-density slots behind)."""
    z_sort = torch.sort(z_vals, 1).values
    z_args = torch.searchsorted(z_sort, z_vals)
    rgbs_sort = torch.zeros_like(rgbs).scatter_(1, z_args[:, :, None].repeat(1, 1, 3), rgbs)
    sig_sort = torch.zeros_like(sigmas).scatter_(1, z_args, sigmas)
    return composite(sig_sort, rgbs_sort, z_sort, white_bkgd=white_bkgd)


def scene_batch_samples(batch_rays: Tensor, diags: Tensor, n_samples: int, jitter: Optional[Tensor] = None,
                        adjust_scale: float = 1.0, shapenet_obj_cood: bool = True):
    """Sample points of one ray batch (Nr, Nb, 8): object-major xyz / viewdir (Nb*Nr, S, 3) for the batched decoder,
    metric depths z (Nr*Nb, S) and the empty-space mask.  Restates scripts/demo.py:528-551."""
    Nr, Nb = batch_rays.shape[:2]
    rays = batch_rays.reshape(-1, 8)
    z_coarse = unit_interval_samples(rays[:, 6:7], rays[:, 7:8], n_samples, jitter)
    empty = z_coarse == -1
    xyz = rays[:, None, :3] + z_coarse[:, :, None] * rays[:, None, 3:6]
    viewdir = rays[:, 3:6].unsqueeze(-2).repeat(1, n_samples, 1)
    d = diags.view(1, -1, 1, 1).repeat(Nr, 1, 1, 1).flatten(0, 1)
    z_vals = torch.norm((xyz - rays[:, None, :3]) * (d / 2), p=2, dim=-1)
    z_vals[empty] = -1
    xyz = xyz.view(Nr, Nb, n_samples, 3).permute(1, 0, 2, 3).flatten(0, 1) * adjust_scale
    viewdir = viewdir.view(Nr, Nb, n_samples, 3).permute(1, 0, 2, 3).flatten(0, 1)
    if shapenet_obj_cood:
        xyz = xyz[:, :, [1, 0, 2]]; xyz[:, :, 0] *= (-1)
        viewdir = viewdir[:, :, [1, 0, 2]]; viewdir[:, :, 0] *= (-1)
    return xyz, viewdir, z_vals, empty


def vis_scene(model, obj_poses, obj_wlh, shapecodes, texturecodes, K, H, W, n_samples, manipulation=(0.0, 0.0, 0.0),
              ray_batch_size=2048, rend_aabb=True, shapenet_obj_cood=True, adjust_scale=1.0, jitters=None):
    """Float canvas (H*W,3) and the uint8 image (H,W,3) of all objects rendered into one view (white background).
    ``jitters``: list with one (Nr*Nb, S) draw per ray batch (the reference draws ``rand_like`` per batch)."""
    all_rays, valid, diags = scene_rays(obj_poses, obj_wlh, K, H, W, manipulation, rend_aabb)
    Nb = obj_poses.shape[0]
    valid_rays = all_rays.view(H * W, -1, 8)[valid, ...]
    out = []
    with torch.no_grad():
        for bi, batch in enumerate(torch.split(valid_rays, ray_batch_size)):
            Nr = batch.shape[0]
            jit = None if jitters is None else jitters[bi]
            xyz, viewdir, z_vals, empty = scene_batch_samples(batch, diags, n_samples, jit, adjust_scale, shapenet_obj_cood)
            sig, rgb = _decode(model, xyz, viewdir, shapecodes, texturecodes)
            rgb = rgb.view(Nb, Nr, n_samples, 3).permute(1, 0, 2, 3).flatten(0, 1)
            sig = sig.view(Nb, Nr, n_samples).permute(1, 0, 2).flatten(0, 1)
            rgb[empty, ...] = 1
            sig[empty] = 0
            out.append(scene_composite(sig.view(-1, Nb * n_samples), rgb.view(-1, Nb * n_samples, 3), z_vals.view(-1, Nb * n_samples))[0])
    canvas = torch.ones(H * W, 3)
    if out:
        canvas[valid, :] = torch.cat(out, 0)
    return canvas, (canvas.view(H, W, 3).numpy() * 255).astype(np.uint8)


# --------------------------------------------------------------------------
# synthetic "nuScenes car" objects -- shared by tests & bench
# --------------------------------------------------------------------------

WLH_MEAN = np.array([1.94, 4.64, 1.71], dtype=np.float32)     # src/optimizer_nuscenes.py:27
WLH_STD = np.array([0.19, 0.46, 0.25], dtype=np.float32)
NUSC_K = np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]], dtype=np.float32)


def synthetic_object(index: int, im_w: int = 1600, im_h: int = 900):
    """A car-sized box at 8-35 m with a plausible camera pose, intrinsics and a
    square roi around the projected centre.  Deterministic in ``index``."""
    rs = np.random.RandomState(1000 + index)
    wlh = (WLH_MEAN + WLH_STD * rs.randn(3)).astype(np.float32)
    diag = np.linalg.norm(wlh).astype(np.float32)
    yaw = rs.uniform(-np.pi, np.pi)
    depth = rs.uniform(8.0, 35.0)
    lateral = rs.uniform(-0.25, 0.25) * depth
    # object pose in the camera frame: x right, y down, z forward; object z up
    c, s = np.cos(yaw), np.sin(yaw)
    R_obj = np.array([[c, -s, 0], [0, 0, -1], [s, c, 0]], dtype=np.float32)
    t_obj = np.array([lateral, 1.2, depth], dtype=np.float32)
    # camera pose in the object frame
    R_c2o = R_obj.T
    t_c2o = -R_c2o @ t_obj
    cam_pose = torch.from_numpy(np.concatenate([R_c2o, t_c2o[:, None]], axis=1).astype(np.float32))
    K = torch.from_numpy(NUSC_K.copy())
    u = NUSC_K[0, 0] * t_obj[0] / t_obj[2] + NUSC_K[0, 2]
    v = NUSC_K[1, 1] * t_obj[1] / t_obj[2] + NUSC_K[1, 2]
    half = int(max(16, 0.36 * NUSC_K[0, 0] * diag / depth))
    x0 = int(np.clip(u - half, 0, im_w - 2 * half - 1))
    y0 = int(np.clip(v - half, 0, im_h - 2 * half - 1))
    roi = torch.tensor([x0, y0, x0 + 2 * half, y0 + 2 * half], dtype=torch.int32)
    return dict(wlh=wlh, obj_diag=diag, cam_pose=cam_pose, K=K, roi=roi)


def synthetic_targets(index: int, im_sz: int):
    """Target crop already at im_sz^2 (so the resize is the identity) with an
    elliptical occupancy mask in {-1, 0, 1}."""
    g = torch.Generator().manual_seed(2000 + index)
    img = torch.rand(im_sz, im_sz, 3, generator=g)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, im_sz), torch.linspace(-1, 1, im_sz), indexing="ij")
    r = (xx / 0.8) ** 2 + (yy / 0.55) ** 2
    mask = torch.where(r < 1.0, torch.ones_like(r), -torch.ones_like(r))
    mask = torch.where((r >= 1.0) & (r < 1.3), torch.zeros_like(r), mask)
    return img, mask[..., None]


# --------------------------------------------------------------------------
# KITTI-side host geometry of the cross-domain loop (src/optimizer_kitti.py:638-651)
# --------------------------------------------------------------------------

def roi_process(roi: Tensor, H: Optional[int] = None, W: Optional[int] = None, roi_margin: int = 0, sq_pad: bool = False) -> Tensor:
    """src/utils.py:1392-1415 on an integer box [xmin, ymin, xmax, ymax], restated with python scalars: margin, square padding
    about the centre (float centre and half size, truncated toward zero when written back into the integer box), clip to
    [0, W-1] x [0, H-1]."""
    integer = not roi.dtype.is_floating_point
    x0, y0, x1, y1 = [v - roi_margin if i < 2 else v + roi_margin for i, v in enumerate(roi.tolist())]
    if sq_pad:
        cx, cy = np.float32(x0 + x1) / np.float32(2), np.float32(y0 + y1) / np.float32(2)
        half = np.float64(max(x1 - x0, y1 - y0)) / 2
        vals = [np.float32(cx - half), np.float32(cy - half), np.float32(cx + half), np.float32(cy + half)]
        x0, y0, x1, y1 = [int(v) if integer else float(v) for v in vals]       # int(): toward zero, like the tensor write
    if H is not None and W is not None:
        x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, W - 1), min(y1, H - 1)
    return torch.tensor([x0, y0, x1, y1], dtype=roi.dtype)


def obj_pose_kitti2nusc(obj_pose: Tensor, obj_h: Tensor) -> Tensor:
    """src/utils.py:1354-1366 without the in-place write: (B,3,4) KITTI-convention object poses -> nuScenes convention.
    Column j of the new rotation: x stays, the new y axis is the old z (left), the new z axis is minus the old y (up); the origin
    moves from the box bottom to the box centre (camera y points down, so T_y decreases by h/2)."""
    R, T = obj_pose[:, :, :3], obj_pose[:, :, 3].clone()
    T[:, 1] = T[:, 1] - obj_h / 2
    R_new = torch.stack([R[:, :, 0], R[:, :, 2], -R[:, :, 1]], dim=-1)
    return torch.cat([R_new, T[:, :, None]], dim=-1)


def eval_curves(saved: dict, max_iter: int):
    """What ``collect_eval_results`` (src/utils.py:786-880) plots from a ``codes+poses.pth`` dict: per-iteration mean PSNR (negative
    values zeroed), lidar-count-weighted depth error, mean rotation error in degrees, mean translation error.  Infinite PSNRs and
    NaN rotation errors are cleared the way the reference does it: it indexes the (objects, iterations) table with ``argwhere``'s
    (row, col) PAIRS, i.e. with both numbers as ROW indices, so an offending entry at (r, c) zeroes objects r and c entirely (and
    raises when c >= number of objects) -- restated as is."""
    ps = np.asarray([np.array(v)[:max_iter] for v in saved["psnr_eval"].values()], dtype=np.float64)
    for r, c in np.argwhere(np.isinf(ps)):
        ps[r] = 0
        ps[c] = 0
    ps[ps < 0] = 0
    de = np.asarray([np.array(v)[:max_iter] for v in saved["depth_err_mean"].values()], dtype=np.float64)
    cnt = np.asarray(list(saved["lidar_pts_cnt"].values()), dtype=np.float64)
    R = np.asarray([torch.stack(v).numpy()[:max_iter] for v in saved["R_eval"].values()], dtype=np.float64)
    for r, c in np.argwhere(np.isnan(R)):
        R[r] = 0
        R[c] = 0
    T = np.asarray([torch.stack(v).numpy()[:max_iter] for v in saved["T_eval"].values()], dtype=np.float64)
    return ps.mean(0), (de * cnt[:, None]).sum(0) / cnt.sum(), R.mean(0) / np.pi * 180, T.mean(0)



# --------------------------------------------------------------------------
# feed-forward pose refinement that fills the loop's pose table
# (src/optimizer_nuscenes.py:451-551; its start_wt_est_pose / PnP branch is cv2 and not restated)
# --------------------------------------------------------------------------

def rotvec_to_matrix(v: Tensor) -> Tensor:
    """(...,3) rotation vector -> (...,3,3), Rodrigues.  Stands in for ``pytorch3d.transforms.axis_angle_to_matrix``
    (src/optimizer_nuscenes.py:539; pytorch3d is not installed and unpinned in the reference: PARITY UNPINNED for this function,
    checked against scipy only)."""
    theta = torch.linalg.norm(v, dim=-1, keepdim=True)
    axis = v / theta.clamp_min(1e-12)
    x, y, z = axis[..., 0], axis[..., 1], axis[..., 2]
    o = torch.zeros_like(x)
    Kx = torch.stack([torch.stack([o, -z, y], -1), torch.stack([z, o, -x], -1), torch.stack([-y, x, o], -1)], -2)
    s, c = torch.sin(theta)[..., None], torch.cos(theta)[..., None]
    eye = torch.eye(3, dtype=v.dtype).expand(Kx.shape)
    return eye + s * Kx + (1 - c) * (Kx @ Kx)


def matrix_to_rotvec(R: Tensor) -> Tensor:
    """(...,3,3) -> (...,3) with angle in [0, pi).  Stands in for ``pytorch3d.transforms.matrix_to_axis_angle``
    (src/optimizer_nuscenes.py:537; PARITY UNPINNED like ``rotvec_to_matrix``); valid away from angle pi."""
    cos = ((R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2] - 1) / 2).clamp(-1, 1)
    ang = torch.acos(cos)
    w = torch.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    s = torch.sin(ang)
    scale = torch.where(s > 1e-6, ang / (2 * s.clamp_min(1e-6)), torch.full_like(s, 0.5))          # (acos gives [0, pi]: sin >= 0)
    return w * scale[..., None]


def roi_normalised_points(pts: Tensor, roi: Tensor) -> Tuple[Tensor, Tensor]:
    """``normalize_by_roi(pts, roi, need_square=True)`` (src/utils.py:1175-1197): (N,2,n) pixel points relative to the roi centre,
    divided by the roi's LONGER side (not half of it).  Returns (points, longer side (N,))."""
    w, h = roi[:, 2] - roi[:, 0], roi[:, 3] - roi[:, 1]
    cx, cy = (roi[:, 2] + roi[:, 0]) / 2, (roi[:, 3] + roi[:, 1]) / 2
    out = pts.clone()
    out[:, 0, :] -= cx.unsqueeze(-1)
    out[:, 1, :] -= cy.unsqueeze(-1)
    dim = torch.maximum(w, h)
    out /= dim.view(-1, 1, 1)
    return out, dim


def pose_head(params: Dict[str, Tensor], im_feat: Tensor, box_uv: Tensor) -> Tensor:
    """``SUPNeRF.pose_update`` (src/model_supnerf.py:226-239) from a state-dict: pose_layer_0.. encode the 16 projected-corner
    coordinates, regress_layer_0.. take [image feature | pose feature], out_delta_layer gives the 6-vector."""
    f, j = box_uv, 0
    while f"pose_layer_{j}.0.weight" in params:
        f = torch.relu(F.linear(f, params[f"pose_layer_{j}.0.weight"], params[f"pose_layer_{j}.0.bias"]))
        j += 1
    d, j = torch.cat([im_feat, f], -1), 0
    while f"regress_layer_{j}.0.weight" in params:
        d = torch.relu(F.linear(d, params[f"regress_layer_{j}.0.weight"], params[f"regress_layer_{j}.0.bias"]))
        j += 1
    return F.linear(d, params["out_delta_layer.weight"], params["out_delta_layer.bias"])


def pose_refine_step(pose_update, im_feat: Tensor, src_pose: Tensor, wlh: Tensor, roi: Tensor, K: Tensor, K_inv: Tensor,
                     to_rotvec=matrix_to_rotvec, to_matrix=rotvec_to_matrix) -> Tensor:
    """``fw_pose_one_step`` (src/optimizer_nuscenes.py:509-551).  ``pose_update(im_feat, uv16)`` is the pose head.  The two
    rotation conversions are arguments so that the fixture generator can run the reference's building blocks around the SAME
    conversions (pytorch3d is absent)."""
    uv = project_points(box_corners(src_pose, wlh), K)                   # :517 corners_of_box_batch + view_points_batch(normalize=True)
    uv_n, dim = roi_normalised_points(uv[:, :2, :], roi)                 # :520
    delta = pose_update(im_feat, uv_n.reshape(im_feat.shape[0], -1)).clone()     # :523-527
    delta[:, :3] *= (torch.pi * 2)                                       # :531
    delta[:, 3:5] *= dim.unsqueeze(-1)                                   # :532
    delta[:, 5] += 1                                                     # :533
    pred_R = to_matrix(to_rotvec(src_pose[:, :, :3]) + delta[:, :3])     # :536-539
    T_src = src_pose[:, :, 3:]
    uvw = torch.matmul(K, T_src)                                         # :542
    pred_u = uvw[:, 0] / uvw[:, 2] + delta[:, 3:4]
    pred_v = uvw[:, 1] / uvw[:, 2] + delta[:, 4:5]
    pred_Z = src_pose[:, 2, 3:] * delta[:, 5:]
    pred_T = torch.matmul(K_inv, torch.cat([pred_u * pred_Z, pred_v * pred_Z, pred_Z], dim=1).unsqueeze(-1))     # :546-547
    return torch.cat([pred_R, pred_T], dim=2)


def pose_refine_table(pose_update, im_feat, src_pose, wlh, roi, K, K_inv, iters=3, **conv) -> Tensor:
    """``fw_pose_update`` (src/optimizer_nuscenes.py:451-507, start_wt_est_pose False): (B, iters+1, 3, 4), the start pose first."""
    table = [src_pose]
    with torch.no_grad():
        for _ in range(iters):
            table.append(pose_refine_step(pose_update, im_feat, table[-1], wlh, roi, K, K_inv, **conv))
    return torch.stack(table, dim=1)
