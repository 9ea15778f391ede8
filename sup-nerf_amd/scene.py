"""Multi-object scene rendering: the counterpart of ``OptimizerDemo.vis_scene`` (scripts/demo.py:425-579).

All reconstructed objects are rendered into one camera view: every pixel gets one ray per object (in that object's
frame, origin divided by diag/2, box entry/exit depths from the slab test, -1 where the object does not cover the
pixel); the samples of all objects are decoded in ONE batched-code decoder launch per ray batch (object-major rays,
``Nb`` codes), merged per pixel by metric depth and composited against white by ``ops.scene_composite`` (HIP).

The method's attributes (``self.obj_poses``, ``self.obj_wlh``, ``self.shapecodes`` ...) are function arguments here;
the ray table is built on the host exactly like the reference does (python loop over a handful of objects).
"""
from typing import Optional, Sequence

import numpy as np
import torch

from . import ops
from . import utils as U
from ._lib import SnrError


def corners_of_box_batch(obj_poses: torch.Tensor, wlh: torch.Tensor) -> torch.Tensor:
    """(Nb,3,8) box corners in the camera frame, nuScenes convention (src/utils.py:1110-1148, is_kitti=False)."""
    sx = torch.tensor([1, 1, 1, 1, -1, -1, -1, -1], dtype=wlh.dtype, device=wlh.device)
    sy = torch.tensor([1, -1, -1, 1, 1, -1, -1, 1], dtype=wlh.dtype, device=wlh.device)
    sz = torch.tensor([1, 1, -1, -1, 1, 1, -1, -1], dtype=wlh.dtype, device=wlh.device)
    local = torch.stack([wlh[:, 1:2] / 2 * sx, wlh[:, 0:1] / 2 * sy, wlh[:, 2:3] / 2 * sz], dim=1)
    return torch.matmul(obj_poses[:, :, :3], local) + obj_poses[:, :, 3:4]


def view_points_batch(points: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """Perspective projection (src/utils.py:1032-1075 with normalize=True): rows (u, v, 1)."""
    uvw = torch.matmul(K, points)
    return uvw / uvw[:, 2:3, :]


def scene_rays(obj_poses, obj_wlh, K, H, W, manipulation=(0.0, 0.0, 0.0), rend_aabb=True):
    """(H,W,Nb,8) ray table [o/(diag/2), dir, near, far], valid-pixel mask (H*W,), diagonals (Nb,) -- scripts/demo.py:437-523.
    CPU tensors, as in the reference."""
    obj_poses, obj_wlh, K = obj_poses.detach().cpu().float(), obj_wlh.detach().cpu().float(), K.detach().cpu().float()
    Nb = obj_poses.shape[0]
    table = torch.full((H, W, Nb, 8), -1.0)
    poses = obj_poses.clone()
    poses[:, :, 3] += torch.tensor(manipulation, dtype=torch.float32).unsqueeze(0)
    uv = view_points_batch(corners_of_box_batch(poses, obj_wlh), K.unsqueeze(0).repeat(Nb, 1, 1))
    rois = torch.stack([uv[:, 0].min(dim=1)[0], uv[:, 1].min(dim=1)[0], uv[:, 0].max(dim=1)[0], uv[:, 1].max(dim=1)[0]], dim=1).type(torch.int32)
    diags = []
    for i in range(Nb):
        x0, y0 = max(int(rois[i, 0]), 0), max(int(rois[i, 1]), 0)                    # roi_process(roi, H, W, 0, False)
        x1, y1 = min(int(rois[i, 2]), W - 1), min(int(rois[i, 3]), H - 1)
        R_c2o = poses[i, :3, :3].transpose(0, 1)
        cam_pose = torch.cat([R_c2o, -R_c2o @ poses[i, :3, 3:4]], dim=1)
        wlh = obj_wlh[i].numpy()
        diag = np.linalg.norm(wlh).astype(np.float32)
        diags.append(diag)
        if x1 <= x0 or y1 <= y0:
            continue
        rays_o, viewdir = U.get_rays(K, cam_pose, [x0, y0, x1, y1])
        table[y0:y1, x0:x1, i, :3] = rays_o.view(y1 - y0, x1 - x0, 3) / (diag / 2)
        table[y0:y1, x0:x1, i, 3:6] = viewdir.view(y1 - y0, x1 - x0, 3)
        if rend_aabb:
            ow, ol, oh = wlh
            half = torch.from_numpy(np.asarray([ol / diag, ow / diag, oh / diag]).reshape(1, 3).repeat(rays_o.shape[0], axis=0))   # float64 like np.asarray
            o_n = torch.from_numpy(rays_o.numpy() / (diag / 2))
            t_near, t_far, hit = U._slab(o_n, viewdir, -half, half)
            near = torch.full((rays_o.shape[0],), -1.0)
            far = torch.full((rays_o.shape[0],), -1.0)
            near[hit] = t_near[hit].float(); far[hit] = t_far[hit].float()
            table[y0:y1, x0:x1, i, 6] = near.view(y1 - y0, x1 - x0)
            table[y0:y1, x0:x1, i, 7] = far.view(y1 - y0, x1 - x0)
        else:
            dist = torch.linalg.norm(cam_pose[:, -1])
            table[y0:y1, x0:x1, i, 6] = (dist - diag / 2) / (diag / 2)
            table[y0:y1, x0:x1, i, 7] = (dist + diag / 2) / (diag / 2)
    diags = torch.tensor(diags, dtype=torch.float32)
    valid = (table[:, :, :, 7].view(H * W, Nb) - table[:, :, :, 6].view(H * W, Nb)).max(-1)[0] > 0
    return table, valid, diags


def render_scene_batch(model, device, batch_rays, diags, shapecodes, texturecodes, n_samples, jitter=None, adjust_scale=1.0,
                       shapenet_obj_cood=True):
    """One ray batch (Nr, Nb, 8) -> rgb (Nr,3), depth (Nr,), acc_trans (Nr,)  (scripts/demo.py:527-566)."""
    dev = torch.device(device)
    Nr, Nb = batch_rays.shape[:2]
    rays = batch_rays.reshape(-1, 8)
    step = 1.0 / n_samples
    t = torch.linspace(0, 1 - step, n_samples)[None, :].repeat(rays.shape[0], 1)
    t = t + (torch.rand_like(t) if jitter is None else jitter.cpu()) * step            # CPU draw, like the reference's CPU ray table
    rays, t = rays.to(dev), t.to(dev)
    z_coarse = rays[:, 6:7] * (1 - t) + rays[:, 7:8] * t
    empty = z_coarse == -1
    xyz = rays[:, None, :3] + z_coarse[:, :, None] * rays[:, None, 3:6]
    d = diags.to(dev).view(1, Nb, 1, 1).repeat(Nr, 1, 1, 1).flatten(0, 1)
    z_vals = torch.norm((xyz - rays[:, None, :3]) * (d / 2), p=2, dim=-1)
    z_vals = torch.where(empty, torch.full_like(z_vals, -1.0), z_vals)
    xyz = xyz.view(Nr, Nb, n_samples, 3).permute(1, 0, 2, 3).flatten(0, 1) * adjust_scale
    viewdir = rays[:, 3:6].view(Nr, Nb, 1, 3).permute(1, 0, 2, 3).expand(Nb, Nr, n_samples, 3).flatten(0, 1)
    if shapenet_obj_cood:
        xyz = torch.stack([-xyz[..., 1], xyz[..., 0], xyz[..., 2]], -1)
        viewdir = torch.stack([-viewdir[..., 1], viewdir[..., 0], viewdir[..., 2]], -1)
    sig, rgb = model(xyz.contiguous(), viewdir.contiguous(), shapecodes.to(dev), texturecodes.to(dev))   # object-major, Nb codes
    rgb = rgb.view(Nb, Nr, n_samples, 3).permute(1, 0, 2, 3).reshape(Nr, Nb * n_samples, 3)
    sig = sig.view(Nb, Nr, n_samples).permute(1, 0, 2).reshape(Nr, Nb * n_samples)
    empty = empty.view(Nr, Nb * n_samples)
    rgb = torch.where(empty[..., None], torch.ones_like(rgb), rgb)                       # empty space: white, zero density
    sig = torch.where(empty, torch.zeros_like(sig), sig)
    return ops.scene_composite(sig, rgb, z_vals.view(Nr, Nb * n_samples), white_bkgd=True, run_length=n_samples)


def vis_scene(model, device, obj_poses, obj_wlh, shapecodes, texturecodes, K, H, W, n_samples, manipulation=(0.0, 0.0, 0.0),
              ray_batch_size=4096, rend_aabb=True, shapenet_obj_cood=True, adjust_scale=1.0, jitters: Optional[Sequence] = None,
              return_float=False):
    """uint8 canvas (H,W,3) with all objects rendered at their (manipulated) poses; white where nothing is hit.
    ``jitters``: optional list of (Nr*Nb, S) draws, one per ray batch (tests); default ``torch.rand_like`` per batch."""
    if obj_poses.shape[0] != shapecodes.shape[0] or obj_poses.shape[0] != texturecodes.shape[0] or obj_poses.shape[0] != obj_wlh.shape[0]:
        raise SnrError("vis_scene: obj_poses, obj_wlh, shapecodes and texturecodes must describe the same number of objects")
    table, valid, diags = scene_rays(obj_poses, obj_wlh, K, H, W, manipulation, rend_aabb)
    valid_rays = table.view(H * W, -1, 8)[valid, ...]
    out = []
    with torch.no_grad():
        for bi, batch in enumerate(torch.split(valid_rays, ray_batch_size)):
            out.append(render_scene_batch(model, device, batch, diags, shapecodes, texturecodes, n_samples,
                                          None if jitters is None else jitters[bi], adjust_scale, shapenet_obj_cood)[0])
    canvas = torch.ones(H * W, 3)
    if out:
        canvas[valid, :] = torch.cat(out, 0).cpu()
    img = (canvas.view(H, W, 3).numpy() * 255).astype(np.uint8)
    return (img, canvas) if return_float else img
