"""Family-A renderer API (the one the reference's optimisers and trainer import from ``utils``):
same function names, argument order and return values as src/utils.py:94-672 of the reference, with
the work done by the HIP kernels.

Differences in mechanism, not in results:
  * everything after ray generation (sampling, /obj_diag, symmetry flip, kitti2nusc, shapenet frame,
    positional encoding, decoder, composite) is ONE kernel launch when ``model`` is a
    ``supnerf_amd.model`` decoder and n_samples divides 128; otherwise three HIP launches
    (encode -> model(...) -> composite);
  * near/far are computed on the device when the pose lives there, so no ``.tolist()`` host sync
    (src/utils.py:468-469) stalls the stream; like the reference they are detached from the pose;
  * random draws come from the same generators in the same order as the reference (CPU ``torch.rand(S)``
    jitter, ``np.random.permutation`` ray subset, ``random.uniform`` symmetry coin), so a seeded run
    consumes identical random numbers.
"""
import contextlib
import functools
import random
import threading

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from ._lib import SnrError
from .ops import Z_PER_OBJECT, Z_PER_RAY, Z_SHARED


# ------------------------------------------------------------------------------------ rays
# Two small caches take the per-call host work out of the optimisers' loops, which call the render functions with the same crop, mask,
# roi and intrinsics in every iteration (src/optimizer_nuscenes.py:716-726): the camera-frame pixel directions (they do not depend on
# the pose) and the resized targets (src/utils.py:447-456 redoes the same bilinear resize per call).  Values are what the uncached
# code computes; entries are keyed by content (K, roi, grid) or by tensor identity + version (crop, mask) and live on the device.
_CAM_CACHE, _TGT_CACHE, _CACHE_MAX = {}, {}, 32
# The render functions are entered from several threads at once (one per GPU under nn.DataParallel, src/trainer_unified_nuscenes.py:227-229;
# one per stream in a caller's own thread pool): reads of these dictionaries are single ``dict.get`` calls (atomic), every
# evict-and-insert runs under this lock, and entries are immutable tuples / tensors keyed by content, so a racing reader sees either the
# old entry or the new one -- both correct.
_CACHE_LOCK = threading.Lock()


def _cache_put(cache, key, value, limit=None):
    with _CACHE_LOCK:
        while len(cache) >= (limit or _CACHE_MAX):
            cache.pop(next(iter(cache)), None)
        cache[key] = value
    return value


def _stream_key(device):
    """Part of every device-cache key: the stream the entry was filled on.  A tensor another thread cached a moment ago on ITS stream may
    not be written yet from the point of view of this thread's stream; entries are therefore per (device, stream) -- a thread on its own
    stream fills its own (one more small launch, once)."""
    d = torch.device(device) if not isinstance(device, torch.device) else device
    if d.type != "cuda":
        return (str(d), 0)
    return (str(d), ops.raw_stream(d))


def _cam_table(K, px, py, like, key=None):
    """[(px - cx)/fx, (py - cy)/fy, 1] as ``like``'s dtype on ``like``'s device (the first half of get_rays, src/utils.py:122-131)."""
    if key is not None:
        Kc = K.detach().cpu() if torch.is_tensor(K) and K.is_cuda else K          # (one host copy, not four scalar reads of a device tensor)
        key = key + (tuple(float(v) for v in (Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2])), _stream_key(like.device), like.dtype)
        hit = _CAM_CACHE.get(key)
        if hit is not None:
            return hit
    cx, cy, fx, fy = K[0, 2], K[1, 2], K[0, 0], K[1, 1]
    cam = torch.stack([(px - cx) / fx, (py - cy) / fy, torch.ones_like(px)], -1).type_as(like)
    return cam if key is None else _cache_put(_CAM_CACHE, key, cam)


def _fusable_pose(c2w):
    """The one-launch ray kernels take a single fp32 (3,4) pose that lives on the GPU.  A (4,4) pose takes the torch formulation: the
    reference's sphere bounds are the norm of the WHOLE last column (``cam_pose[:, -1]``, src/utils.py:468), which for a homogeneous pose
    includes the 1 -- the kernel computes |t| of the three translation entries only."""
    return torch.is_tensor(c2w) and c2w.is_cuda and tuple(c2w.shape) == (3, 4) and c2w.dtype == torch.float32


def _pixel_dirs(K, c2w, px, py, key=None):
    cam = _cam_table(K, px, py, c2w, key)
    if _fusable_pose(c2w) and cam.numel() > 0:
        # pose on the GPU: rotate + normalise + broadcast the origin in ONE launch (backward: one launch), instead of ~6 + ~10 torch launches
        rays_o, viewdir, _ = ops.CamRays.apply(c2w[:3, :].unsqueeze(0), cam.reshape(1, -1, 3), None, None, 0)
        return rays_o, viewdir
    world = (cam[..., None, :] * c2w[..., :3, :3]).sum(-1)
    unit = world / torch.norm(world, dim=-1, keepdim=True)
    origin = c2w[..., :3, -1].expand(world.shape)
    return origin.reshape(-1, 3), unit.reshape(-1, 3)


def get_rays(K, c2w, roi, uv_steps=None):
    """src/utils.py:107-135: rays through a pixel grid over roi=[xmin,ymin,xmax,ymax], row-major (y,x);
    returns (rays_o (N,3), viewdirs (N,3)) on c2w's device, differentiable wrt c2w."""
    x0, y0, x1, y1 = [int(v) for v in roi]
    nx, ny = (int(uv_steps[0]), int(uv_steps[1])) if uv_steps is not None else (x1 - x0, y1 - y0)
    xs = torch.linspace(x0, x1 - 1, nx)
    ys = torch.linspace(y0, y1 - 1, ny)
    return _pixel_dirs(K, c2w, xs[None, :].expand(ny, nx), ys[:, None].expand(ny, nx), key=("grid", x0, y0, x1, y1, nx, ny))


def get_rays_specified(K, c2w, x_vec, y_vec):
    """src/utils.py:138-151: rays through listed pixel coordinates (numpy integer arrays)."""
    return _pixel_dirs(K, c2w, torch.from_numpy(np.asarray(x_vec)), torch.from_numpy(np.asarray(y_vec)))


def get_rays_srn(H, W, focal, c2w):
    """src/utils.py:94-104 (ShapeNet-SRN camera convention; unused by the shipped drivers)."""
    xs, ys = torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H)
    px, py = xs[None, :].expand(H, W), ys[:, None].expand(H, W)
    cam = torch.stack([(px - W * .5) / focal, -(py - H * .5) / focal, -torch.ones_like(px)], -1).type_as(c2w)
    world = (cam[..., None, :] * c2w[..., :3, :3]).sum(-1)
    unit = world / torch.norm(world, dim=-1, keepdim=True)
    return c2w[..., :3, -1].expand(world.shape).reshape(-1, 3), unit.reshape(-1, 3)


# ------------------------------------------------------------------------------------ sampling
def _sphere_bounds(cam_pose, obj_diag):
    """near/far = |camera centre| -/+ diag/2, detached (src/utils.py:468-469).  CPU pose: python floats exactly
    as the reference; GPU pose: 0-dim device tensors, no host sync."""
    if cam_pose.is_cuda:
        dist = cam_pose[:, -1].detach().float().norm()
        half = float(obj_diag) / 2
        return dist - half, dist + half
    dist = np.linalg.norm(cam_pose[:, -1].tolist())
    return dist - obj_diag / 2, dist + obj_diag / 2


def _linspace(start, end, steps, device):
    """torch.linspace for python-float or 0-dim-tensor endpoints (same two-sided fp32 formula)."""
    if not torch.is_tensor(start):
        return torch.linspace(start, end, steps).to(device)
    start, end = start.float(), end.float()
    i = torch.arange(steps, device=start.device, dtype=torch.float32)
    step = (end - start) / max(steps - 1, 1)
    lo = start + step * i
    hi = end - step * (steps - 1 - i)
    return torch.where(i < steps // 2, lo, hi)


# Test hook: injected jitter instead of the generator draws (parity tests inject the reference's numbers).  A tensor replaces every
# following draw, a list is consumed one tensor per draw.  ``jitter_override(...)`` is the re-entrant form: it is THREAD-LOCAL (two
# threads rendering concurrently each see their own injection, or none) and restores on exit.  The module attribute ``JITTER_OVERRIDE`` is
# the older process-wide form, kept for single-threaded test code; the thread-local value wins.
JITTER_OVERRIDE = None
_TLS = threading.local()
_NOT_SET = object()


@contextlib.contextmanager
def jitter_override(value):
    """``with utils.jitter_override(t): ...`` -- inject jitter for the calls of THIS thread inside the block."""
    prev = getattr(_TLS, "jitter", _NOT_SET)
    _TLS.jitter = value
    try:
        yield
    finally:
        _TLS.jitter = prev


def _jitter_override():
    v = getattr(_TLS, "jitter", _NOT_SET)
    if v is _NOT_SET:
        v = JITTER_OVERRIDE
    if isinstance(v, list):
        with _CACHE_LOCK:
            return v.pop(0) if v else None
    return v


def _shared_depths(near, far, n_samples, device, z_fixed=False, jitter=None):
    """The (S,) depth vector of ``sample_from_rays`` (src/utils.py:159-164).  ``jitter`` overrides the
    ``torch.rand(S)`` draw (tests)."""
    if z_fixed:
        return _linspace(near, far, n_samples, device)
    half = (far - near) / (2 * n_samples)
    z = _linspace(near + half, far - half, n_samples, device)
    if jitter is None:
        jitter = _jitter_override()
        if jitter is None:
            jitter = torch.rand(n_samples)                   # same CPU generator, same numbers (uploaded from pageable memory, see _draw_jitter)
    return z + jitter.to(device, non_blocking=True) * (far - near) / (2 * n_samples)


def sample_from_rays(ro, vd, near, far, N_samples, z_fixed=False):
    """src/utils.py:154-167: xyz (N,S,3), viewdir repeated (N,S,3), z_vals (S,).  HIP encode kernel."""
    z = _shared_depths(near, far, N_samples, ro.device, z_fixed)
    cfg = ops.RenderCfg(N_samples, Z_SHARED, max(ro.shape[0], 1), 0, 0)
    if ro.requires_grad or vd.requires_grad:
        zz = z.type_as(ro)
        return ro.unsqueeze(-2) + vd.unsqueeze(-2) * zz.unsqueeze(-1), vd.unsqueeze(-2).repeat(1, N_samples, 1), z
    xyz, vdir, _ = ops.encode(ro, vd, z, torch.ones(1, device=ro.device), None, cfg)
    return xyz, vdir, z


def _unit_depths(near, far, n_samples, jitter=None):
    """src/renderer.py:27-41 == src/utils.py:170-184: stratified depths between per-ray near/far (N,1); device ``rand_like``
    jitter (one draw of (N,S) from the device generator, like the reference)."""
    step = 1.0 / n_samples
    t = torch.linspace(0, 1 - step, n_samples, device=near.device)[None, :].repeat(near.shape[0], 1)
    if jitter is None:
        jitter = _jitter_override()
    t = t + (torch.rand_like(t) if jitter is None else jitter.to(t.device)) * step
    return near * (1 - t) + far * t


def sample_from_rays_v2(rays, n_samples):
    """src/utils.py:170-184 (imported by name in scripts/demo.py:14): rays (B,8) = [origin, direction, near, far] ->
    stratified depths (B, n_samples).  The module-level twin of ``NeRFRenderer.sample_from_ray``."""
    return _unit_depths(rays[:, -2:-1], rays[:, -1:], n_samples)


def _frame(sym_flip, kitti2nusc, shapenet_obj_cood):
    return _frame_cached(bool(sym_flip), bool(kitti2nusc), bool(shapenet_obj_cood))


@functools.lru_cache(maxsize=None)
def _frame_cached(sym_flip, kitti2nusc, shapenet_obj_cood):
    """Row-major 3x3 combining, in the reference's order (src/utils.py:475-495): y mirror, KITTI->nuScenes
    (x,y,z)->(x,z,-y), nuScenes->ShapeNet (x,y,z)->(-y,x,z)."""
    m = np.eye(3, dtype=np.float32)
    if sym_flip:
        m = np.diag([1.0, -1.0, 1.0]).astype(np.float32) @ m
    if kitti2nusc:
        m = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float32) @ m
    if shapenet_obj_cood:
        m = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=np.float32) @ m
    return tuple(m.reshape(-1).tolist())


def _sym_coin(sym_aug):
    return bool(sym_aug) and random.uniform(0, 1) > 0.5


_CONST_CACHE = {}


def _const(value, n, device):
    """(n,) fp32 device tensor filled with ``value``, cached (object sizes do not change between the calls of a loop)."""
    key = (float(value), int(n), _stream_key(device))
    t = _CONST_CACHE.get(key)
    if t is None:
        t = _cache_put(_CONST_CACHE, key, torch.full((int(n),), float(value), device=device), limit=256)
    return t


def clear_caches():
    """Drop the cached camera tables, resized targets and per-object constants (they pin device memory)."""
    with _CACHE_LOCK:
        _CAM_CACHE.clear(); _TGT_CACHE.clear(); _CONST_CACHE.clear()


def _draw_jitter(n_samples, device):
    """The ``torch.rand(S)`` draw of ``sample_from_rays`` (src/utils.py:162) from the same CPU generator, uploaded without a host sync.
    From PAGEABLE memory on purpose: the 256 bytes go through the runtime's staging buffer (~10 us, no stream sync).  A pinned
    allocation per call (round 2) is not reusable until its copy has run, so a caller that queues calls faster than the GPU finishes
    them -- any no_grad loop -- paid a fresh hipHostMalloc (~1 ms) per call until the allocator's cache had grown to the queue depth."""
    jitter = _jitter_override()
    if jitter is None:
        jitter = torch.rand(n_samples)
    return jitter.to(device, non_blocking=True)


def _rays_and_depths(K, cam_pose, roi, uv_steps, obj_diag, n_samples, ids=None, pixels=None):
    """Rays of the roi's pixel grid (optionally the subset ``ids``), or of the listed ``pixels`` = (x_vec, y_vec) in image coordinates, and
    the shared depth vector: what render_rays{,_v2,_specified} / render_full_img compute before the render (src/utils.py:445,462-470,
    515-520 + sample_from_rays :159-164).  Pose on the GPU: ONE launch (camera table cached, rotation + normalisation + sphere bounds +
    two-sided linspace + jitter in ``snr_cam_rays_fwd``, backward one launch)."""
    if _fusable_pose(cam_pose):
        if pixels is not None:
            x_vec, y_vec = np.asarray(pixels[0]), np.asarray(pixels[1])
            cam = _cam_table(K, torch.from_numpy(x_vec), torch.from_numpy(y_vec), cam_pose,
                             key=("pix", x_vec.shape, hash(x_vec.tobytes()), hash(y_vec.tobytes()))).reshape(-1, 3)
        else:
            x0, y0, x1, y1 = [int(v) for v in roi]
            nx, ny = (int(uv_steps[0]), int(uv_steps[1])) if uv_steps is not None else (x1 - x0, y1 - y0)
            xs, ys = torch.linspace(x0, x1 - 1, nx), torch.linspace(y0, y1 - 1, ny)
            cam = _cam_table(K, xs[None, :].expand(ny, nx), ys[:, None].expand(ny, nx), cam_pose, key=("grid", x0, y0, x1, y1, nx, ny)).reshape(-1, 3)
        if ids is not None:
            cam = cam[torch.as_tensor(ids, device=cam.device)]
        if cam.shape[0] > 0:
            dev = cam_pose.device
            rays_o, viewdir, z = ops.CamRays.apply(cam_pose[:3, :].unsqueeze(0), cam.unsqueeze(0), _const(float(obj_diag) / 2, 1, dev),
                                                   _draw_jitter(n_samples, dev).reshape(1, n_samples), n_samples)
            return rays_o, viewdir, z[0]
    if pixels is not None:
        rays_o, viewdir = get_rays_specified(K, cam_pose, pixels[0], pixels[1])
    else:
        rays_o, viewdir = get_rays(K, cam_pose, roi, uv_steps=uv_steps)
    if ids is not None:
        rays_o, viewdir = rays_o[ids], viewdir[ids]
    near, far = _sphere_bounds(cam_pose, obj_diag)
    return rays_o, viewdir, _shared_depths(near, far, n_samples, rays_o.device)


# ------------------------------------------------------------------------------------ composite
def _composite(sigmas, rgbs, z_vals, z_mode, white_bkgd, rays_per_obj=0):
    sig = sigmas.squeeze(-1) if sigmas.dim() == rgbs.dim() else sigmas
    return ops.Composite.apply(sig, rgbs, z_vals, z_mode, white_bkgd, rays_per_obj)


def volume_rendering2(sigmas, rgbs, z_vals):
    """src/utils.py:202-217: sigmas (N,S,1), rgbs (N,S,3), z_vals (S,) -> rgb (N,3), depth (N,), acc_trans (N,)."""
    return _composite(sigmas, rgbs, z_vals, Z_SHARED, False)


def volume_rendering(sigmas, rgbs, z_vals):
    """src/utils.py:187-199 (legacy, two outputs; the reference omits the relu here, which only matters for
    negative densities -- the decoder's softplus never produces them)."""
    rgb, depth, _ = _composite(sigmas, rgbs, z_vals, Z_SHARED, False)
    return rgb, depth


def volume_rendering_batch(sigmas, rgbs, z_vals):
    """src/utils.py:220-233: sigmas (B,n,S,1), rgbs (B,n,S,3), z_vals (B,S) -> (B,n,3), (B,n), (B,n)."""
    B, n, S = rgbs.shape[:3]
    rgb, depth, acc = _composite(sigmas.reshape(B * n, S), rgbs.reshape(B * n, S, 3), z_vals, Z_PER_OBJECT, False, rays_per_obj=n)
    return rgb.view(B, n, 3), depth.view(B, n), acc.view(B, n)


# ------------------------------------------------------------------------------------ targets
def _resize_to(img, mask_occ, im_sz, device):
    """``_resize`` + the move to ``device`` as (n,3) / (n,1) ray targets, cached per (crop, mask, size, device): the optimisers pass the
    same CPU crop in every iteration."""
    key = (img.data_ptr(), img._version, tuple(img.shape), mask_occ.data_ptr(), mask_occ._version, tuple(mask_occ.shape), int(im_sz), _stream_key(device))
    hit = _TGT_CACHE.get(key)
    # the cached targets are handed out as they are (a clone per call would be a launch per call): a caller that edits them in place --
    # reference-style code does, e.g. ``occ_pixels[occ_pixels < 0] = 0`` -- bumps their version counter, and the entry is rebuilt
    if hit is not None and hit[0]() is img and hit[1]() is mask_occ and hit[2]._version == hit[4] and hit[3]._version == hit[5]:
        return hit[2], hit[3]
    im, mk = _resize(img, mask_occ, im_sz)
    tgt, occ = im.reshape(-1, 3).to(device), mk.reshape(-1, 1).to(device)
    import weakref
    _cache_put(_TGT_CACHE, key, (weakref.ref(img), weakref.ref(mask_occ), tgt, occ, tgt._version, occ._version))
    return tgt, occ


def _pixel_targets(img, mask_occ, x_vec, y_vec, device):
    """``img[y_vec, x_vec, :].to(device)``, ``mask_occ[y_vec, x_vec, :].to(device)`` (src/utils.py:521-522), cached per (crop, mask, pixel
    list, device) like ``_resize_to``: the optimisers ask for the same lidar pixels of the same CPU crop in every iteration, and the two
    gathers + uploads were a fifth of the call's host time."""
    if not (isinstance(img, torch.Tensor) and isinstance(mask_occ, torch.Tensor) and not img.is_cuda and not mask_occ.is_cuda):
        return img[y_vec, x_vec, :].to(device), mask_occ[y_vec, x_vec, :].to(device)
    xa, ya = np.asarray(x_vec), np.asarray(y_vec)
    key = ("pix", img.data_ptr(), img._version, tuple(img.shape), mask_occ.data_ptr(), mask_occ._version, tuple(mask_occ.shape),
           xa.shape, hash(xa.tobytes()), hash(ya.tobytes()), _stream_key(device))
    hit = _TGT_CACHE.get(key)
    if hit is not None and hit[0]() is img and hit[1]() is mask_occ and hit[2]._version == hit[4] and hit[3]._version == hit[5]:
        return hit[2], hit[3]
    tgt, occ = img[y_vec, x_vec, :].to(device), mask_occ[y_vec, x_vec, :].to(device)
    import weakref
    _cache_put(_TGT_CACHE, key, (weakref.ref(img), weakref.ref(mask_occ), tgt, occ, tgt._version, occ._version))
    return tgt, occ


def _resize(img, mask_occ, im_sz):
    """Bilinear, no antialias (torchvision 0.13 tensor ``Resize``); mask truncated through int32
    (src/utils.py:447-456)."""
    im = F.interpolate(img.permute(2, 0, 1)[None], size=(im_sz, im_sz), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    mk = F.interpolate(mask_occ.permute(2, 0, 1)[None], size=(im_sz, im_sz), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    return im, mk.type(torch.int32).type(torch.float32)


# ------------------------------------------------------------------------------------ render core
def _is_native(model):
    return hasattr(model, "fused_render") and hasattr(model, "packed_weights")


def _render_shared_z(model, device, rays_o, viewdir, z, obj_diag, frame, shapecode, texturecode):
    """Family-A tail: points = frame * ((o + z d) / obj_diag), black background, z shared by all rays."""
    S = z.shape[0]
    dev = torch.device(device)
    rays_o, viewdir, z = rays_o.to(dev), viewdir.to(dev), z.to(dev)
    B = shapecode.shape[0]
    div = _const(obj_diag, B, dev)
    cfg = ops.RenderCfg(S, Z_SHARED, rays_o.shape[0] // B, getattr(model, "shape_blocks", 0), getattr(model, "texture_blocks", 0),
                        frame=frame, precision=None)
    if rays_o.shape[0] == 0:
        e = torch.empty(0, device=dev)
        return e.view(0, 3), e, e
    if _is_native(model) and ops.fused_supported(S):
        return model.fused_render(rays_o, viewdir, z, div, None, shapecode, texturecode, cfg)
    # unfused: HIP encode -> caller's decoder -> HIP composite
    if rays_o.requires_grad or viewdir.requires_grad:
        m = torch.tensor(frame, device=dev).view(3, 3)
        xyz = ((rays_o[:, None, :] + viewdir[:, None, :] * z[None, :, None]) / float(obj_diag)) @ m.T
        vdir = (viewdir @ m.T)[:, None, :].repeat(1, S, 1)
    else:
        xyz, vdir, _ = ops.encode(rays_o, viewdir, z, div, None, cfg)
    sigmas, rgbs = model(xyz, vdir, shapecode, texturecode)
    return volume_rendering2(sigmas, rgbs, z)


def render_rays(model, device, img, mask_occ, cam_pose, obj_diag, K, roi, n_samples, shapecode, texturecode, shapenet_obj_cood,
                sym_aug, kitti2nusc=False, n_rays=2500):
    """src/utils.py:380-432: a random subset of n_rays pixels of the full roi."""
    n_all = int(roi[2] - roi[0]) * int(roi[3] - roi[1])
    n_rays = int(np.minimum(n_all, n_rays))
    ids = np.random.permutation(n_all)[:n_rays]
    rays_o, viewdir, z = _rays_and_depths(K, cam_pose, roi, None, obj_diag, n_samples, ids=ids)
    rgb_tgt = img.reshape(-1, 3)[ids].to(device)
    occ_pixels = mask_occ.reshape(-1, 1)[ids].to(device)
    frame = _frame(_sym_coin(sym_aug), kitti2nusc, shapenet_obj_cood)
    rgb, depth, acc = _render_shared_z(model, device, rays_o, viewdir, z, obj_diag, frame, shapecode, texturecode)
    return rgb, depth, acc, rgb_tgt, occ_pixels


def render_rays_v2(model, device, img, mask_occ, cam_pose, obj_diag, K, roi, n_samples, shapecode, texturecode, shapenet_obj_cood,
                   sym_aug, kitti2nusc=False, im_sz=64, n_rays=None):
    """src/utils.py:435-502: im_sz x im_sz grid over the roi (the optimisers' render call)."""
    rgb_tgt, occ_pixels = _resize_to(img, mask_occ, im_sz, device)
    ids = None
    if n_rays is not None:
        n_all = int(im_sz) * int(im_sz)
        n_rays = int(np.minimum(n_all, n_rays))
        ids = np.random.permutation(n_all)[:n_rays]
        rgb_tgt, occ_pixels = rgb_tgt[ids], occ_pixels[ids]
    rays_o, viewdir, z = _rays_and_depths(K, cam_pose, roi, [im_sz, im_sz], obj_diag, n_samples, ids=ids)
    frame = _frame(_sym_coin(sym_aug), kitti2nusc, shapenet_obj_cood)
    rgb, depth, acc = _render_shared_z(model, device, rays_o, viewdir, z, obj_diag, frame, shapecode, texturecode)
    return rgb, depth, acc, rgb_tgt, occ_pixels


def render_rays_specified(model, device, img, mask_occ, cam_pose, obj_diag, K, roi, x_vec, y_vec, n_samples, shapecode, texturecode,
                          shapenet_obj_cood, sym_aug, kitti2nusc=False):
    """src/utils.py:504-551: rays at listed pixels of the crop (lidar pixels in the optimisers)."""
    rays_o, viewdir, z = _rays_and_depths(K, cam_pose, roi, None, obj_diag, n_samples, pixels=(x_vec + int(roi[0]), y_vec + int(roi[1])))
    rgb_tgt, occ_pixels = _pixel_targets(img, mask_occ, x_vec, y_vec, device)
    frame = _frame(_sym_coin(sym_aug), kitti2nusc, shapenet_obj_cood)
    rgb, depth, acc = _render_shared_z(model, device, rays_o, viewdir, z, obj_diag, frame, shapecode, texturecode)
    return rgb, depth, acc, rgb_tgt, occ_pixels


def prepare_pixel_samples(img, mask_occ, cam_pose, obj_diag, K, roi, n_rays, n_samples, shapenet_obj_cood, sym_aug, im_sz=None):
    """src/utils.py:330-377: pre-sampled ray batch for the trainer: xyz (n,S,3), viewdir (n,S,3), z_vals (S,),
    rgb_tgt (n,3), occ_pixels (n,1).  Stays on the inputs' device (the datasets call it on CPU workers; there the
    arithmetic is plain torch, on the GPU it is the encode kernel)."""
    near, far = _sphere_bounds(cam_pose, obj_diag)
    if im_sz is None:
        rays_o, viewdir = get_rays(K, cam_pose, roi)
    else:
        rays_o, viewdir = get_rays(K, cam_pose, roi, uv_steps=[im_sz, im_sz])
        img, mask_occ = _resize(img, mask_occ, im_sz)
    n_rays = int(np.minimum(rays_o.shape[0], n_rays))
    ids = np.random.permutation(rays_o.shape[0])[:n_rays]
    rays_o, viewdir = rays_o[ids], viewdir[ids]
    rgb_tgt = img.reshape(-1, 3)[ids]
    occ_pixels = mask_occ.reshape(-1, 1)[ids]
    z = _shared_depths(near, far, n_samples, rays_o.device)
    frame = _frame(_sym_coin(sym_aug), False, shapenet_obj_cood)
    if rays_o.is_cuda:
        cfg = ops.RenderCfg(n_samples, Z_SHARED, max(rays_o.shape[0], 1), 0, 0, frame=frame)
        xyz, vdir, _ = ops.encode(rays_o, viewdir, z, torch.full((1,), float(obj_diag), device=rays_o.device), None, cfg)
    else:
        m = torch.tensor(frame).view(3, 3)
        xyz = ((rays_o[:, None, :] + viewdir[:, None, :] * z[None, :, None]) / obj_diag) @ m.T
        vdir = (viewdir @ m.T)[:, None, :].repeat(1, n_samples, 1)
    return xyz, vdir, z, rgb_tgt, occ_pixels


def render_full_img(model, device, cam_pose, obj_sz, K, roi, n_samples, shapecode, texturecode, shapenet_obj_cood, out_depth=False,
                    debug_occ=False, kitti2nusc=False):
    """src/utils.py:554-616: every pixel of the roi; returns (H,W,3) [and (H,W) depth].  The reference renders in
    slabs of max(roi_w, roi_h) rays to bound activation memory; the fused kernel keeps activations in registers, so
    the whole roi is one launch."""
    obj_diag = np.linalg.norm(obj_sz).astype(np.float32)
    frame = _frame(False, kitti2nusc, shapenet_obj_cood)
    with torch.no_grad():
        rays_o, viewdir, z = _rays_and_depths(K, cam_pose, roi, None, obj_diag, n_samples)
        rgb, depth, acc = _render_shared_z(model, device, rays_o, viewdir, z, obj_diag, frame, shapecode, texturecode)
    h, w = int(roi[3] - roi[1]), int(roi[2] - roi[0])
    if debug_occ:
        raise SnrError("debug_occ opens a cv2 window in the reference; not available in this package")
    if out_depth:
        return rgb.reshape(h, w, 3), depth.reshape(h, w)
    return rgb.reshape(h, w, 3)


def render_virtual_imgs(model, device, obj_sz, K, n_samples, shapecode, texturecode, shapenet_obj_cood, radius=40., tilt=np.pi / 6,
                        pan_num=8, img_sz=128, kitti2nusc=False):
    """src/utils.py:619-672: pan_num turntable views at fixed radius / tilt.  Returns the list of (img_sz,img_sz,3)
    CPU tensors; the reference additionally draws the object axes with cv2 arrows (visualisation only, omitted)."""
    cx, cy = float(K[0, 2]), float(K[1, 2])
    roi = np.asarray([cx - img_sz / 2, cy - img_sz / 2, cx + img_sz / 2, cy + img_sz / 2]).astype(np.int64)
    cam_init = np.asarray([[0, 0, 1, -radius], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 1]]).astype(np.float32)
    ct, st = np.cos(tilt), np.sin(tilt)
    cam_tilt = np.asarray([[ct, 0, st, 0], [0, 1, 0, 0], [-st, 0, ct, 0], [0, 0, 0, 1]]).astype(np.float32) @ cam_init
    views = []
    for pan in np.linspace(0, 2 * np.pi, pan_num, endpoint=False):
        cp, sp = np.cos(pan), np.sin(pan)
        pose = np.asarray([[cp, -sp, 0, 0], [sp, cp, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]).astype(np.float32) @ cam_tilt
        img = render_full_img(model, device, torch.from_numpy(pose[:3, :]), obj_sz, K, roi, n_samples, shapecode, texturecode,
                              shapenet_obj_cood, kitti2nusc=kitti2nusc)
        views.append(img.cpu())
    return views


# ------------------------------------------------------------------------------------ box test (used by renderer.py)
def ray_box_intersection_tensor(ray_o, ray_d, aabb_min=None, aabb_max=None):
    """src/utils.py:283-327: slab test; returns (z_in, z_out) of the HIT rays and the boolean hit map, or
    (None, None, None) for an empty ray set.  NaN-propagating min/max like the reference."""
    if aabb_min is None:
        aabb_min = torch.full_like(ray_o, -1.)
    if aabb_max is None:
        aabb_max = torch.full_like(ray_o, 1.)
    t_near, t_far, hit = _slab(ray_o, ray_d, aabb_min, aabb_max)
    if hit.shape[0] == 0:
        return None, None, None
    return t_near[hit], t_far[hit], hit


def _slab(ray_o, ray_d, aabb_min, aabb_max):
    inv = torch.reciprocal(ray_d)
    ta, tb = (aabb_min - ray_o) * inv, (aabb_max - ray_o) * inv
    lo, hi = torch.minimum(ta, tb), torch.maximum(ta, tb)
    t_near = torch.maximum(torch.maximum(lo[..., 0], lo[..., 1]), lo[..., 2])
    t_far = torch.minimum(torch.minimum(hi[..., 0], hi[..., 1]), hi[..., 2])
    hit = t_far > t_near
    hit = torch.logical_and(hit, (t_far * hit) > 0)
    return t_near, t_far, hit


def ray_box_intersection(ray_o, ray_d, aabb_min=None, aabb_max=None):
    """src/utils.py:236-280: numpy twin of the slab test."""
    if aabb_min is None:
        aabb_min = np.ones_like(ray_o) * -1.
    if aabb_max is None:
        aabb_max = np.ones_like(ray_o)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.reciprocal(ray_d)
        ta, tb = (aabb_min - ray_o) * inv, (aabb_max - ray_o) * inv
    lo, hi = np.minimum(ta, tb), np.maximum(ta, tb)
    t_near = np.maximum(np.maximum(lo[..., 0], lo[..., 1]), lo[..., 2])
    t_far = np.minimum(np.minimum(hi[..., 0], hi[..., 1]), hi[..., 2])
    hit = t_far > t_near
    hit = np.logical_and(hit, (t_far * hit) > 0)
    if hit.shape[0] == 0:
        return None, None, None
    return t_near[hit], t_far[hit], hit


# ------------------------------------------------------------------------------------ dataset-side geometry on the loop's path
def roi_process(roi, H=None, W=None, roi_margin=0, sq_pad=False):
    """src/utils.py:1392-1415: grow the 2D box [xmin, ymin, xmax, ymax] by ``roi_margin``, optionally pad the shorter side to a
    square about the centre, then clip to the image (truncated objects end up non-square).  Integer tensors stay integer: the
    centre and half size are computed in the tensor's dtype exactly like the reference's in-place writes (float results are
    truncated toward zero on assignment)."""
    out = roi.clone()
    out[0:2] -= roi_margin
    out[2:4] += roi_margin
    if sq_pad:
        cx, cy = (out[0] + out[2]) / 2, (out[1] + out[3]) / 2
        sz = np.maximum(out[2] - out[0], out[3] - out[1])
        out[0], out[2] = cx - sz / 2, cx + sz / 2
        out[1], out[3] = cy - sz / 2, cy + sz / 2
    if H is not None and W is not None:
        out[0:2] = torch.maximum(out[0:2], torch.as_tensor(0))
        out[2] = torch.minimum(out[2], torch.as_tensor(W - 1))
        out[3] = torch.minimum(out[3], torch.as_tensor(H - 1))
    return out


_KITTI2NUSC_RX = np.array([[1., 0., 0.], [0., 0., -1.], [0., 1., 0.]], dtype=np.float32)


def obj_pose_kitti2nusc(obj_pose_src, obj_h):
    """src/utils.py:1354-1366, called by src/optimizer_kitti.py:638-639: KITTI object poses (B,3,4) (x front, y down, z left, origin
    at the box bottom) -> the nuScenes convention the renderer is trained in: the origin moves up by h/2 (T_y -= h/2) and the axes
    turn by R <- R R_x.  Like the reference it writes the shifted translation back into ``obj_pose_src`` (its ``pose_T`` is a view)."""
    pose_T = obj_pose_src[:, :, 3:]
    pose_T[:, 1, 0] -= (obj_h / 2)
    R_x = torch.from_numpy(_KITTI2NUSC_RX).unsqueeze(0).repeat(obj_pose_src.shape[0], 1, 1)
    return torch.cat([torch.matmul(obj_pose_src[:, :, :3], R_x), pose_T], dim=-1)


def rot_dist(R1, R2):
    """src/utils.py:713-722: geodesic angle between rotations (B,3,3) -> (B,)."""
    d = torch.matmul(R1, torch.transpose(R2, -1, -2))
    trace = d.diagonal(dim1=-2, dim2=-1).sum(-1)
    return torch.acos((trace.clamp(-1, 3) - 1) / 2)


def calc_pose_err(est_poses, tgt_poses):
    """src/utils.py:675-683: rotation (rad) and translation (m) errors of (B,3,4) poses."""
    err_R = rot_dist(est_poses[:, :3, :3], tgt_poses[:, :3, :3])
    err_T = torch.sqrt(torch.sum((est_poses[:, :3, 3] - tgt_poses[:, :3, 3]) ** 2, dim=-1))
    return err_R, err_T
