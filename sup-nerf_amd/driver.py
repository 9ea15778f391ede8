"""Test-time optimisation loop on the HIP render path: the package's own counterpart of
``OptimizerNuScenes.optimize_objs_w_pose_unified`` (src/optimizer_nuscenes.py:553-794 of the reference; KITTI /
Waymo twins src/optimizer_kitti.py:606-885), plus object sharding across GPUs.

Per object and iteration, exactly as the reference: pose -> cam2opt (:685-699), ``render_rays_v2`` (:716-726),
``loss = loss_rgb + loss_occ_coef * loss_occ`` (:729-736), backward, foreground PSNR (:739-744), pose errors
(:747-750), depth error at "lidar" pixels via ``render_rays_specified`` under no_grad (:757-765), AdamW step over four
parameter groups (:1762-1769) skipped for the first ``reg_iters + 1`` iterations (:768-769), learning rates halved
every ``lr_half_interval`` by re-creating AdamW (:1771-1775, which resets the Adam moments -- kept).

Objects are independent, so multi-GPU = one process per GPU, each taking a contiguous slice of the object list
(the reference's ``--num_subset/--id_subset`` slicing, src/data_nuscenes.py:318-320, but the ``L mod N`` tail is not
dropped) with no communication inside the loop; one ``all_gather`` of the per-object metric rows at the end (RCCL
when the backend is nccl, gloo on CPU in the tests).

No dataset is available offline: objects come from ``synthetic.py``.  The rotation parametrisation
(pytorch3d ``axis_angle_to_matrix`` / ``matrix_to_axis_angle`` in the reference, un-pinned and not installed) is
restated here via Rodrigues' formula; parity for it is pinned only by self-consistency tests.
"""
import json
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import synthetic, utils as U

DEFAULT_HPAMS = {     # the keys of jsonfiles/supnerf.nusc.vehicle.car.json that the loop reads
    "n_samples": 64, "render_im_sz": 32, "roi_margin": 5, "shapenet_obj_cood": 1, "sym_aug": 0, "loss_occ_coef": 0.1,
    "dataset": {"name": "nusc", "img_h": 900, "img_w": 1600, "mask_pixels": 2500, "max_dist": 40},
    "net_hyperparams": {"shape_blocks": 3, "texture_blocks": 1, "latent_dim": 256, "num_xyz_freq": 10, "num_dir_freq": 4},
    "optimize": {"num_opts": 100, "opt_cam_pose": 0, "lr_shape": 0.02, "lr_texture": 0.02, "lr_pose": 0.01, "lr_half_interval": 1000},
}


# jsonfiles/supnerf.kitti.car.json differs from the nuScenes file on the loop's path only in these (tests/golden/kitti.json holds
# the reference file's values)
KITTI_OVERRIDES = {"roi_margin": 15, "dataset": {"name": "kitti", "mask_pixels": 1600, "max_dist": 40, "min_depth": 3}}


# jsonfiles/supnerf.waymo.car.json: optimize_waymo.py runs the KITTI-convention loop (labels converted by obj_pose_kitti2nusc,
# src/optimizer_waymo.py:125,140: roi_margin 15, sq_pad) on 1920 x 1280 images (tests/golden/waymo.json holds the reference file's values)
WAYMO_OVERRIDES = {"roi_margin": 15, "dataset": {"name": "waymo", "mask_pixels": 2500, "max_dist": 40, "min_depth": 3, "min_lidar_cnt": 10}}


def load_hpams(path: Optional[str] = None, dataset: str = "nusc") -> dict:
    """Read a reference config file (jsonfiles/*.json) -- same keys, verbatim -- or the shipped defaults of ``dataset``
    ('nusc' | 'kitti' | 'waymo')."""
    if path is None:
        hp = json.loads(json.dumps(DEFAULT_HPAMS))
        if dataset == "kitti":
            hp.update(json.loads(json.dumps(KITTI_OVERRIDES)))
        elif dataset == "waymo":
            hp.update(json.loads(json.dumps(WAYMO_OVERRIDES)))
        elif dataset != "nusc":
            raise ValueError(f"unknown dataset {dataset!r}")
        return hp
    with open(path) as f:
        return json.load(f)


# ------------------------------------------------------------------ rotations (restated, quaternion-free Rodrigues)
def axis_angle_to_matrix(v: torch.Tensor) -> torch.Tensor:
    """(...,3) rotation vector -> (...,3,3).  R = I + sin(t)/t K + (1-cos t)/t^2 K^2, series near t = 0."""
    t2 = (v * v).sum(-1, keepdim=True)
    t = torch.sqrt(t2.clamp_min(1e-24))
    small = t2 < 1e-8
    a = torch.where(small, 1 - t2 / 6, torch.sin(t) / t)
    b = torch.where(small, 0.5 - t2 / 24, (1 - torch.cos(t)) / t2.clamp_min(1e-24))
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    zero = torch.zeros_like(x)
    K = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], -1).reshape(*v.shape[:-1], 3, 3)
    eye = torch.eye(3, dtype=v.dtype, device=v.device).expand(K.shape)
    return eye + a[..., None] * K + b[..., None] * (K @ K)


def matrix_to_axis_angle(R: torch.Tensor) -> torch.Tensor:
    """(...,3,3) -> (...,3) rotation vector with angle in [0, pi]."""
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    cos = ((tr - 1) / 2).clamp(-1, 1)
    ang = torch.acos(cos)
    w = torch.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    s = torch.sin(ang)
    k = torch.where(s.abs() > 1e-6, ang / (2 * torch.where(s.abs() > 1e-6, s, torch.ones_like(s))), torch.full_like(s, 0.5))
    out = w * k[..., None]
    # near pi the antisymmetric part vanishes: take the axis from the diagonal of (R + I)/2
    near_pi = cos < -0.999
    if near_pi.any():
        d = ((torch.diagonal(R, dim1=-2, dim2=-1) + 1) / 2).clamp_min(0).sqrt()
        sign = torch.sign(w)
        sign = torch.where(sign == 0, torch.ones_like(sign), sign)
        out = torch.where(near_pi[..., None], d * sign * ang[..., None], out)
    return out


def rot_dist(R1: torch.Tensor, R2: torch.Tensor) -> torch.Tensor:
    """Geodesic angle between rotations (src/utils.py:713-722)."""
    d = R1 @ R2.transpose(-1, -2)
    tr = (d[..., 0, 0] + d[..., 1, 1] + d[..., 2, 2]).clamp(-1, 3)
    return torch.acos(((tr - 1) / 2).clamp(-1, 1))


# ------------------------------------------------------------------ sharding (src/data_nuscenes.py:318-320, tail kept)
def shard_slice(n_items: int, world_size: int, rank: int) -> range:
    """Contiguous slice of rank ``rank``; the first ``n_items % world_size`` ranks take one extra item."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def gather_metric_rows(rows: torch.Tensor, ids: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """all_gather of per-object metric rows (n_local, n_cols) -> (n_items, n_cols) ordered by object id, on every
    rank.  The only collective of the optimise path."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        out = rows.new_zeros(n_items, rows.shape[1])
        out[ids.long()] = rows
        return out
    world = dist.get_world_size(group)
    n_max = (n_items + world - 1) // world
    pad = rows.new_full((n_max, rows.shape[1] + 1), -1.0)
    pad[: rows.shape[0], 0] = ids.to(rows.dtype)
    pad[: rows.shape[0], 1:] = rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = rows.new_zeros(n_items, rows.shape[1])
    for b in bufs:
        keep = b[:, 0] >= 0
        out[b[keep, 0].long()] = b[keep, 1:]
    return out


# ------------------------------------------------------------------ the stock pose head in the loop (src/optimizer_nuscenes.py:451-551)
def _box_corners(obj_pose, wlh):
    """(B,3,8) box corners in the camera frame: ``corners_of_box_batch(obj_pose, wlh)`` (src/utils.py:1110-1148, nuScenes order: x
    forward, y left, z up)."""
    sx = torch.tensor([1, 1, 1, 1, -1, -1, -1, -1], dtype=wlh.dtype, device=wlh.device)
    sy = torch.tensor([1, -1, -1, 1, 1, -1, -1, 1], dtype=wlh.dtype, device=wlh.device)
    sz = torch.tensor([1, 1, -1, -1, 1, 1, -1, -1], dtype=wlh.dtype, device=wlh.device)
    local = torch.stack([wlh[:, 1:2] / 2 * sx, wlh[:, 0:1] / 2 * sy, wlh[:, 2:3] / 2 * sz], dim=1)
    return torch.matmul(obj_pose[:, :, :3], local) + obj_pose[:, :, 3:4]


def fw_pose_one_step(model, im_feat, src_pose, wlh, roi, K, K_inv):
    """One step of the feed-forward pose refiner, ``OptimizerNuScenes.fw_pose_one_step`` (src/optimizer_nuscenes.py:509-551): project
    the current box (``corners_of_box_batch`` + ``view_points_batch(normalize=True)``), normalise the 8 corners by the roi
    (``normalize_by_roi(need_square=True)``, src/utils.py:1175-1197), let the STOCK pose head ``model.pose_update`` (plain PyTorch,
    src/model_supnerf.py:226-239) regress a 6-vector, and apply it: rotation vector += 2 pi d[:3]; projected centre += d[3:5] * roi size;
    depth *= (d[5] + 1).  (B,3,4) object poses in, (B,3,4) out.  Rotation conversions: Rodrigues (``axis_angle_to_matrix`` above; the
    reference calls pytorch3d, un-pinned)."""
    uvw = torch.matmul(K, _box_corners(src_pose, wlh))
    uv = (uvw / uvw[:, 2:3, :])[:, :2, :]
    w_, h_ = roi[:, 2] - roi[:, 0], roi[:, 3] - roi[:, 1]
    cx, cy = (roi[:, 2] + roi[:, 0]) / 2, (roi[:, 3] + roi[:, 1]) / 2
    dim = torch.maximum(w_, h_)
    uv_n = torch.stack([uv[:, 0, :] - cx.unsqueeze(-1), uv[:, 1, :] - cy.unsqueeze(-1)], dim=1) / dim.view(-1, 1, 1)
    B = im_feat.shape[0]
    d = model.pose_update(im_feat, uv_n.reshape(B, -1))
    d_rot, d_uv, d_z = d[:, :3] * (math.pi * 2), d[:, 3:5] * dim.unsqueeze(-1), d[:, 5:] + 1
    pred_R = axis_angle_to_matrix(matrix_to_axis_angle(src_pose[:, :, :3]) + d_rot)
    c = torch.matmul(K, src_pose[:, :, 3:])
    pred_u = c[:, 0] / c[:, 2] + d_uv[:, 0:1]
    pred_v = c[:, 1] / c[:, 2] + d_uv[:, 1:2]
    pred_Z = src_pose[:, 2, 3:] * d_z
    pred_T = torch.matmul(K_inv, torch.cat([pred_u * pred_Z, pred_v * pred_Z, pred_Z], dim=1).unsqueeze(-1))
    return torch.cat([pred_R, pred_T], dim=2)


def fw_pose_update(model, im_feat, src_pose, wlh, roi, K, K_inv, iters=3):
    """``OptimizerNuScenes.fw_pose_update`` (src/optimizer_nuscenes.py:451-507) without its PnP start (``start_wt_est_pose``: cv2): the
    start pose followed by ``iters`` refinement steps -> the pose table (B, iters + 1, 3, 4) that the loop's first ``reg_iters + 1``
    iterations render at (:641-651,684-689); the optimiser's ``rot_vec`` / ``trans_vec`` start from its last entry (:652,664-666).
    All of it stock PyTorch on the inputs' device, under ``no_grad`` like the caller (:641)."""
    with torch.no_grad():
        table = [src_pose]
        for _ in range(iters):
            table.append(fw_pose_one_step(model, im_feat, table[-1], wlh, roi, K, K_inv))
        return torch.stack(table, dim=1)


# ------------------------------------------------------------------ one object
def make_optimizer(shapecode, texturecode, rot_vec, trans_vec, lr):
    """AdamW over four groups (src/optimizer_nuscenes.py:1762-1769)."""
    return torch.optim.AdamW([{"params": shapecode, "lr": lr["lr_shape"]}, {"params": texturecode, "lr": lr["lr_texture"]},
                              {"params": rot_vec, "lr": lr["lr_pose"]}, {"params": trans_vec, "lr": lr["lr_pose"]}])


def losses(rgb_rays, acc_trans_rays, rgb_tgt, occ_pixels, loss_occ_coef):
    """src/optimizer_nuscenes.py:729-744: (loss, foreground mse used for the PSNR log)."""
    a = torch.abs(occ_pixels)
    denom = a.sum() + 1e-9
    loss_rgb = (((rgb_rays - rgb_tgt) ** 2) * a).sum() / denom
    loss_occ = (torch.exp(-occ_pixels * (0.5 - acc_trans_rays.unsqueeze(-1))) * a).sum() / denom
    fg = occ_pixels.clamp_min(0)
    mse_fg = (((rgb_rays - rgb_tgt) ** 2) * fg).sum() / (fg.sum() + 1e-9)
    return loss_rgb + loss_occ_coef * loss_occ, mse_fg


def optimize_object(model, device, obj: Dict, hpams: dict, shapecode0, texturecode0, pose_noise=(0.05, 0.3), reg_iters=3,
                    n_lidar=64, seed=0, log=None, jitter=None, info: Optional[dict] = None, pose_per_iter=None):
    """Optimise codes and object pose of one object against its (synthetic) target.  Returns a metric tensor
    (num_opts, 4) = [psnr, depth_err, rot_err, trans_err] per iteration, and the final codes / pose.

    With a native decoder this is the fused iteration of ``optimize_objects_batched`` at one object (about thirty launches, no host
    round trip), fed with the depth jitter the reference's loop would have drawn: two ``torch.rand(S)`` per iteration from the global
    CPU generator, in order.  ``optimize_object_api`` is the same loop written against the public functions, call for call like the
    reference (needed for ``sym_aug``, which flips a python coin inside every render call, for foreign decoders and for ``log``).

    ``pose_per_iter`` ((reg_iters + 1, 3, 4) OBJECT poses in the camera frame, e.g. from ``fw_pose_update``; None = the noise-perturbed
    ground truth): the loop's first ``reg_iters + 1`` iterations render at these poses and the optimiser's ``rot_vec`` / ``trans_vec`` start
    from the last one (src/optimizer_nuscenes.py:641-668,684-689)."""
    if pose_per_iter is not None:
        pose_per_iter = torch.as_tensor(pose_per_iter, dtype=torch.float32).reshape(-1, 3, 4)
        if pose_per_iter.shape[0] != reg_iters + 1:
            raise U.SnrError(f"pose_per_iter holds {pose_per_iter.shape[0]} poses, the loop renders reg_iters + 1 = {reg_iters + 1} of them")
    if not U._is_native(model) or hpams.get("sym_aug", 0) or log is not None or not U.ops.fused_supported(hpams["n_samples"]):
        return optimize_object_api(model, device, obj, hpams, shapecode0, texturecode0, pose_noise, reg_iters, n_lidar, seed, log, jitter, info,
                                   pose_per_iter=pose_per_iter)
    T, S = hpams["optimize"]["num_opts"], hpams["n_samples"]
    if jitter is None:
        jitter = torch.stack([torch.stack([torch.rand(S), torch.rand(S)]) for _ in range(T)]) if T else torch.zeros(0, 2, S)
    m, sc, tc, pose = optimize_objects_batched(model, device, [obj], hpams, shapecode0, texturecode0, [seed], pose_noise, reg_iters, n_lidar,
                                               jitter=jitter[:, :, None, :], info=info,
                                               pose_per_iter=None if pose_per_iter is None else pose_per_iter[None])
    return m[0].cpu(), sc, tc, pose[0]


def optimize_object_api(model, device, obj: Dict, hpams: dict, shapecode0, texturecode0, pose_noise=(0.05, 0.3), reg_iters=3,
                        n_lidar=64, seed=0, log=None, jitter=None, info: Optional[dict] = None, pose_per_iter=None):
    """``optimize_object`` through the public render API, one call per reference call (src/optimizer_nuscenes.py:674-783): ~350 launches
    per iteration, host-bound at one object."""
    opt = hpams["optimize"]
    S, im_sz = hpams["n_samples"], hpams["render_im_sz"]
    dev = torch.device(device)
    rs = np.random.RandomState(seed)
    K, roi, obj_diag = obj["K"], obj["roi"], obj["obj_diag"]
    img, mask = obj["img"], obj["mask"]
    # ground-truth OBJECT pose in the camera frame and a perturbed start
    R_c2o, t_c2o = obj["cam_pose"][:, :3], obj["cam_pose"][:, 3:]
    R_gt = R_c2o.T
    t_gt = (-R_gt @ t_c2o)
    gt_pose = torch.cat([R_gt, t_gt], -1)
    rot_vec = (matrix_to_axis_angle(R_gt[None]) + torch.from_numpy(rs.randn(1, 3).astype(np.float32)) * pose_noise[0]).to(dev)
    trans_vec = (t_gt.T + torch.from_numpy(rs.randn(1, 3).astype(np.float32)) * pose_noise[1]).to(dev)
    if pose_per_iter is not None:          # the pose head's table: start from its last pose (src/optimizer_nuscenes.py:652,664-666)
        pose_per_iter = torch.as_tensor(pose_per_iter, dtype=torch.float32).reshape(-1, 3, 4).to(dev)
        rot_vec = matrix_to_axis_angle(pose_per_iter[-1, :3, :3][None]).contiguous()
        trans_vec = pose_per_iter[-1, :3, 3][None].clone()
    rot_vec.requires_grad_(); trans_vec.requires_grad_()
    shapecode = shapecode0.detach().clone().to(dev).requires_grad_()
    texturecode = texturecode0.detach().clone().to(dev).requires_grad_()
    lr = {k: opt[k] for k in ("lr_shape", "lr_texture", "lr_pose")}
    optim = make_optimizer(shapecode, texturecode, rot_vec, trans_vec, lr)
    # the depth pixels: the object's lidar returns with their measured depths (the reference's metric, :751-765), or -- synthetic objects
    # without a depth map -- random foreground pixels, the metric then being the change of rendered depth against the first iteration
    x_vec, y_vec, gt_depth = _lidar_pixels(obj, rs, n_lidar)
    if info is not None:
        info["lidar_count"] = torch.tensor([len(x_vec)], dtype=torch.int32)
    metrics = torch.zeros(opt["num_opts"], 4, device=dev)      # stays on the device: the reference's per-iteration .item() logging
    gt_dev = gt_pose.to(dev)                                    # would put three host syncs into every iteration
    depth0 = None if gt_depth is None else torch.from_numpy(gt_depth).to(dev)
    for it in range(opt["num_opts"]):
        optim.zero_grad()
        if jitter is not None:                                   # (num_opts, 2, S): the two draws of this iteration (tests)
            U.JITTER_OVERRIDE = jitter[it, 0]
        if pose_per_iter is not None and it <= reg_iters:        # "the first a few to load pre-computed poses" (:684-689)
            R, t = pose_per_iter[it, :3, :3], pose_per_iter[it, :3, 3:]
        else:
            R = axis_angle_to_matrix(rot_vec[0])
            t = trans_vec[0].unsqueeze(-1)
        if not opt.get("opt_cam_pose", 0):                       # object pose is optimised: invert to camera-in-object
            Rc = R.transpose(-2, -1)
            cam2opt = torch.cat([Rc, -Rc @ t], -1)
        else:
            cam2opt = torch.cat([R, t], -1)
        rgb, depth, acc, rgb_tgt, occ = U.render_rays_v2(model, dev, img, mask, cam2opt, obj_diag, K, roi, S, shapecode, texturecode,
                                                         hpams["shapenet_obj_cood"], hpams["sym_aug"], im_sz=im_sz, n_rays=None)
        loss, mse_fg = losses(rgb, acc, rgb_tgt, occ, hpams["loss_occ_coef"])
        loss.backward()
        with torch.no_grad():
            if jitter is not None:
                U.JITTER_OVERRIDE = jitter[it, 1]
            _, d_vec, _, _, _ = U.render_rays_specified(model, dev, img, mask, cam2opt.detach(), obj_diag, K, roi, x_vec, y_vec, S,
                                                        shapecode, texturecode, hpams["shapenet_obj_cood"], hpams["sym_aug"])
            if depth0 is None:
                depth0 = d_vec.clone()
            pred_R = cam2opt[:, :3].detach().T if not opt.get("opt_cam_pose", 0) else cam2opt[:, :3].detach()
            pred_t = (-pred_R @ cam2opt[:, 3:].detach()) if not opt.get("opt_cam_pose", 0) else cam2opt[:, 3:].detach()
            row = torch.stack([-10 * torch.log10(mse_fg.detach()), (d_vec - depth0).abs().sum() / (len(x_vec) + 1e-8),     # (log_eval_depth_v2, :1736-1741)
                               rot_dist(pred_R, gt_dev[:, :3]), (pred_t - gt_dev[:, 3:]).norm()])
        metrics[it] = row
        if it > reg_iters:
            optim.step()
        if (it + 1) % opt["lr_half_interval"] == 0:
            halvings = (it + 1) // opt["lr_half_interval"]
            lr = {k: v * 2 ** (-halvings) for k, v in lr.items()}     # cumulative like update_learning_rate (:1771-1775)
            optim = make_optimizer(shapecode, texturecode, rot_vec, trans_vec, lr)
        if log is not None:
            log(it, float(loss), metrics[it].cpu())
    if jitter is not None:
        U.JITTER_OVERRIDE = None
    return metrics.cpu(), shapecode.detach(), texturecode.detach(), cam2opt.detach()


# ------------------------------------------------------------------ many objects per launch (BASELINE config 3)
def optimize_objects_batched(model, device, objs: List[Dict], hpams: dict, shapecodes0, texturecodes0, seeds: Sequence[int],
                             pose_noise=(0.05, 0.3), reg_iters=3, n_lidar=64, jitter=None, info: Optional[dict] = None, pose_per_iter=None):
    """The iteration of ``optimize_object`` for B objects at once: ONE fused forward, one backward and one depth render of the lidar pixels
    per iteration for all of them (per-object codes, poses, depth tables and targets; the loss is the sum of the per-object losses, so
    every object sees exactly its own gradient), AdamW over the stacked leaves in one launch, metrics kept on the device until the end --
    no host round trip inside the loop (``_optimize_fused``: ~24 launches per iteration for any B).  Jitter: ``jitter``
    (num_opts, 2, B, S) or, by default, drawn up front from one CPU generator per object (seeded like the per-object loop seeds its
    RandomState).  ``info`` (optional dict) receives ``lidar_count`` (B,): the number of depth pixels behind every object's depth metric.
    ``pose_per_iter`` (B, reg_iters + 1, 3, 4): every object's pose table for the render-only iterations, see ``optimize_object``.
    Returns metrics (B, num_opts, 4), shape codes, texture codes, poses (B,3,4).
    (Round 2 also carried a HIP-graph replay of the torch-op iteration; it lost to this eager fused loop, 2.3 vs 1.3 ms per iteration, and
    is gone.)"""
    dev = torch.device(device)
    if hpams.get("sym_aug", 0):
        raise U.SnrError("optimize_objects_batched: sym_aug draws one python coin per object and iteration; use optimize_object")
    if pose_per_iter is not None:
        pose_per_iter = torch.as_tensor(pose_per_iter, dtype=torch.float32)
        if tuple(pose_per_iter.shape) != (len(objs), reg_iters + 1, 3, 4):
            raise U.SnrError(f"pose_per_iter must be (B, reg_iters + 1, 3, 4) = ({len(objs)}, {reg_iters + 1}, 3, 4), got {tuple(pose_per_iter.shape)}")
    return _optimize_fused(model, dev, objs, hpams, shapecodes0, texturecodes0, seeds, pose_noise, reg_iters, n_lidar, jitter, info, pose_per_iter)


def _lidar_pixels(ob, rs, n_lidar):
    """The pixels of the crop whose depth the loop logs every iteration, and their measured depths when the object carries them.
    With ``ob["lidar_xy"]`` (n,2 integer crop pixels x, y) and ``ob["lidar_depth"]`` (n,): the reference's metric -- every lidar return
    on the foreground mask, depth L1 against the measurement (src/optimizer_nuscenes.py:751-765,1736-1741).  Without (synthetic objects
    with no depth map): ``n_lidar`` random foreground pixels, and the metric is the change of rendered depth against iteration 0."""
    if ob.get("lidar_xy") is not None:
        xy = np.asarray(ob["lidar_xy"]).reshape(-1, 2).astype(np.int64)
        depth = np.asarray(ob["lidar_depth"], dtype=np.float32).reshape(-1)
        if depth.shape[0] != xy.shape[0]:
            raise U.SnrError("object: lidar_xy (n,2) and lidar_depth (n,) disagree")
        return xy[:, 0], xy[:, 1], depth
    ys, xs = np.where(ob["mask"][:, :, 0].numpy() > 0)
    pick = rs.permutation(len(ys))[:n_lidar]
    return xs[pick], ys[pick], None


def _loop_inputs(objs, seeds, hpams, pose_noise, n_lidar, dev, pose_per_iter=None):
    """Per-object constants of the loop, built once on the host and moved to the device: perturbed start pose, ground truth, the pixel
    direction tables [(px-cx)/fx, (py-cy)/fy, 1] of the render grid and of the lidar pixels (every object keeps ITS OWN count: the tables
    are padded to the largest, the metric kernel averages each object's first ``lid_cnt`` entries), measured depths when the objects
    carry them, resized targets (the reference resizes the same crop again in every iteration, src/utils.py:447-456)."""
    im_sz = hpams["render_im_sz"]
    rot0, tr0, gtR, gtT, cam, lid, lid_d, tgt, occ = [], [], [], [], [], [], [], [], []
    for ob, seed in zip(objs, seeds):
        rs = np.random.RandomState(seed)
        R_gt = ob["cam_pose"][:, :3].T
        t_gt = -R_gt @ ob["cam_pose"][:, 3:]
        rot0.append(matrix_to_axis_angle(R_gt[None]) + torch.from_numpy(rs.randn(1, 3).astype(np.float32)) * pose_noise[0])
        tr0.append(t_gt.T + torch.from_numpy(rs.randn(1, 3).astype(np.float32)) * pose_noise[1])
        if pose_per_iter is not None:      # (the draws above still happen: the lidar pixels below come from the same RandomState)
            last = pose_per_iter[len(rot0) - 1, -1]
            rot0[-1], tr0[-1] = matrix_to_axis_angle(last[:3, :3][None]), last[:3, 3][None].clone()
        gtR.append(R_gt); gtT.append(t_gt.reshape(3))
        x_vec, y_vec, depth = _lidar_pixels(ob, rs, n_lidar)
        x0, y0, x1, y1 = [int(v) for v in ob["roi"]]
        K = ob["K"]
        cx, cy, fx, fy = K[0, 2], K[1, 2], K[0, 0], K[1, 1]
        gx, gy = torch.linspace(x0, x1 - 1, im_sz), torch.linspace(y0, y1 - 1, im_sz)
        px, py = gx[None, :].expand(im_sz, im_sz).reshape(-1), gy[:, None].expand(im_sz, im_sz).reshape(-1)
        cam.append(torch.stack([(px - cx) / fx, (py - cy) / fy, torch.ones_like(px)], -1))
        lx, ly = torch.from_numpy(x_vec + x0), torch.from_numpy(y_vec + y0)         # integer pixels, like get_rays_specified
        lid.append(torch.stack([(lx - cx) / fx, (ly - cy) / fy, torch.ones_like(lx, dtype=torch.float32)], -1).float().reshape(-1, 3))
        lid_d.append(None if depth is None else torch.from_numpy(depth))
        im, mk = U._resize(ob["img"], ob["mask"], im_sz)
        tgt.append(im.reshape(-1, 3)); occ.append(mk.reshape(-1))
    has_depth = [d is not None for d in lid_d]
    if any(has_depth) and not all(has_depth):
        raise U.SnrError("optimize_objects_batched: either every object of a batch carries lidar_xy / lidar_depth or none does")
    cnt = [v.shape[0] for v in lid]
    n_l = max(cnt) if cnt else 0
    if n_l == 0:
        raise U.SnrError("optimize_objects_batched: no object has a depth pixel (empty foreground / no lidar return)")
    one_ray = torch.tensor([[0.0, 0.0, 1.0]])

    def pad(v, width):          # (the padding rays are rendered and ignored: a valid direction, depth 0)
        return torch.cat([v, (one_ray if width == 3 else torch.zeros(1)).expand(n_l - v.shape[0], *([3] if width == 3 else []))]) if v.shape[0] < n_l else v
    st = lambda xs_: torch.stack(xs_).to(dev).contiguous()
    return dict(rot0=torch.cat(rot0).to(dev), tr0=torch.cat(tr0).to(dev), gtR=st(gtR), gtT=st(gtT), cam=st(cam), lid=st([pad(v, 3) for v in lid]),
                lid_depth=st([pad(d, 1) for d in lid_d]) if all(has_depth) else None, lid_cnt=torch.tensor(cnt, dtype=torch.int32, device=dev),
                tgt=st(tgt), occ=st(occ), diag=torch.tensor([float(ob["obj_diag"]) for ob in objs], device=dev), n_lidar=n_l)


def _optimize_fused(model, dev, objs, hpams, shapecodes0, texturecodes0, seeds, pose_noise, reg_iters, n_lidar, jitter, info=None,
                    pose_per_iter=None):
    """The iteration as ~30 launches for any number of objects: pose -> rays + depths (one launch), the per-object layers (two GEMMs),
    fused render, loss tail (one launch), backward = their four backward launches, the 64-pixel depth render, the metric row (one
    launch), AdamW over the four parameter groups (one launch).  Nothing reads back until the loop has finished."""
    ops = U.ops
    opt = hpams["optimize"]
    S, im_sz, T = hpams["n_samples"], hpams["render_im_sz"], opt["num_opts"]
    B, n = len(objs), im_sz * im_sz
    c = _loop_inputs(objs, seeds, hpams, pose_noise, n_lidar, dev, pose_per_iter)
    n_l = c["n_lidar"]
    rot_vec, trans_vec = c["rot0"].requires_grad_(), c["tr0"].requires_grad_()
    shapecode = shapecodes0.detach().clone().to(dev).contiguous().requires_grad_()
    texturecode = texturecodes0.detach().clone().to(dev).contiguous().requires_grad_()
    if jitter is None:
        gens = [torch.Generator().manual_seed(int(s_)) for s_ in seeds]
        jitter = torch.stack([torch.rand(T, 2, S, generator=g) for g in gens], dim=2)
    jitter = jitter.to(dev).contiguous()
    lr = {k: float(opt[k]) for k in ("lr_shape", "lr_texture", "lr_pose")}
    optim = ops.DeviceAdamW([(shapecode, lr["lr_shape"]), (texturecode, lr["lr_texture"]), (rot_vec, lr["lr_pose"]), (trans_vec, lr["lr_pose"])])
    frame = U._frame(False, False, hpams["shapenet_obj_cood"])
    sb, tb = model.shape_blocks, model.texture_blocks
    half = (c["diag"] / 2).contiguous()
    opt_cam = int(bool(opt.get("opt_cam_pose", 0)))
    coef = float(hpams["loss_occ_coef"])
    metrics = torch.zeros(T, B, 4, device=dev)
    measured = c["lid_depth"] is not None
    depth0 = c["lid_depth"] if measured else torch.zeros(B, n_l, device=dev)       # measured depths, or the rendered depths of iteration 0
    if info is not None:
        info["lidar_count"] = c["lid_cnt"].clone()
    ones = torch.ones(B, device=dev)
    pose = torch.zeros(B, 3, 4, device=dev)
    cfg = ops.RenderCfg(S, ops.Z_PER_OBJECT, n, sb, tb, frame=frame, precision=model.precision)
    cfg_l = ops.RenderCfg(S, ops.Z_PER_OBJECT, n_l, sb, tb, frame=frame, precision=model.precision)
    packed = model.packed_weights()
    table = None
    if pose_per_iter is not None:
        # the render-only iterations' camera poses, made once: cam2opt = [R^T | -R^T t] of the table's object poses unless the camera pose
        # itself is what is optimised (src/optimizer_nuscenes.py:684-699); (reg_iters + 1, B, 3, 4) on the device
        P = pose_per_iter.to(dev).permute(1, 0, 2, 3)
        if not opt_cam:
            Rt = P[..., :3].transpose(-1, -2)
            P = torch.cat([Rt, -Rt @ P[..., 3:]], dim=-1)
        table = P.contiguous()
    frozen = [p for p in model.parameters() if p.requires_grad]     # the decoder is a constant of this loop (the reference leaves its
    for p in frozen:                                                 # weights trainable and pays for unused weight gradients)
        p.requires_grad_(False)
    try:
        for it in range(T):
            from_table = table is not None and it <= reg_iters
            if from_table:      # a pre-computed pose per object: rays straight from the (3,4) camera poses (snr_cam_rays_fwd), nothing to differentiate
                rays_o, viewdir, z = ops.CamRays.apply(table[it], c["cam"], half, jitter[it, 0], S)
            else:
                cam2opt, rays_o, viewdir, z = ops.PoseRays.apply(rot_vec, trans_vec, c["cam"], half, jitter[it, 0], S, opt_cam)
            lat = model.latent_terms(shapecode, texturecode)
            cfg.latent_bias = cfg_l.latent_bias = model.latent_biases(lat)
            if it == 0:         # which arithmetic the loop runs in: "auto" is decided (and range-checked) once, on the first iteration's batch
                def probe(p_):      # (one object of the batch: the decoder is the same for all of them)
                    c1 = ops.copy.copy(cfg)
                    c1.latent_bias = None if cfg.latent_bias is None else cfg.latent_bias[:1]
                    return ops.render_probe(rays_o.detach()[:n], viewdir.detach()[:n], z[:1], c["diag"][:1], None, lat.detach()[:1], packed, c1, p_)
                prec = model._auto_precision(model.precision, n * S, probe)
                cfg.precision = prec
                cfg_l.precision = prec if (prec != "bf16x3" or ops.split_supported(sb, tb, n_l * S)) else "fp32"
            if it > reg_iters:
                rgb, depth, acc = ops.FusedRender.apply(rays_o, viewdir, z, c["diag"], None, lat, packed, cfg)
                loss, lm = ops.LossTail.apply(rgb, acc, c["tgt"], c["occ"], coef, n)
                torch.autograd.backward(loss, ones)
            else:               # a render-only iteration: its gradients are cleared unread (:676,768-769), so nothing is recorded -- the forward then
                with torch.no_grad():      # does not save ReLU bits either (the launch that saves them is 9 % slower and writes 208 B per point)
                    rgb, depth, acc = ops.FusedRender.apply(rays_o, viewdir, z, c["diag"], None, lat, packed, cfg)
                    loss, lm = ops.LossTail.apply(rgb, acc, c["tgt"], c["occ"], coef, n)
            with torch.no_grad():
                if from_table:
                    c2o = table[it]
                    lo, lv, lz = ops.CamRays.apply(c2o, c["lid"], half, jitter[it, 1], S)
                else:
                    c2o, lo, lv, lz = ops.PoseRays.apply(rot_vec.detach(), trans_vec.detach(), c["lid"], half, jitter[it, 1], S, opt_cam)
                d_vec = ops.render_fwd(lo, lv, lz, c["diag"], None, lat.detach(), packed, cfg_l)[1]
                out4 = torch.cat([loss.detach()[:, None], lm], dim=1)
                ops.metric_row(out4, d_vec.view(B, n_l), depth0, it == 0 and not measured, c2o, c["gtR"], c["gtT"], opt_cam, metrics[it],
                               lidar_count=c["lid_cnt"])
                if it == T - 1:
                    pose.copy_(c2o)
            if it > reg_iters:
                optim.step()
            optim.zero_grad()
            if (it + 1) % opt["lr_half_interval"] == 0:
                optim.restart(2.0 ** (-((it + 1) // opt["lr_half_interval"])))
    finally:
        for p in frozen:
            p.requires_grad_(True)
    return metrics.permute(1, 0, 2).contiguous(), shapecode.detach(), texturecode.detach(), pose


def make_objects(ids: Sequence[int], im_sz: int, lidar: bool = False) -> List[Dict]:
    """Synthetic nuScenes-like objects.  ``lidar``: also a synthetic set of lidar returns on the foreground (``lidar_xy`` (n,2) crop pixels,
    ``lidar_depth`` (n,) metres; n differs from object to object like real sweeps), which switches the loop's depth metric to the reference's."""
    out = []
    for i in ids:
        ob = synthetic.synthetic_object(i)
        img, mask = synthetic.synthetic_targets(i, im_sz)
        # the reference whitens the background of the crop (src/optimizer_nuscenes.py:711-713)
        img = img * (mask > 0) + (mask <= 0)
        ob.update(img=img, mask=mask, index=i)
        if lidar:
            xy, depth = synthetic.synthetic_lidar(i, mask, ob)
            ob.update(lidar_xy=xy, lidar_depth=depth)
        out.append(ob)
    return out


def make_kitti_objects(ids: Sequence[int], hpams: dict) -> List[Dict]:
    """BASELINE config 4 (src/optimizer_kitti.py:606-672): synthetic KITTI labels -> the dicts ``optimize_object`` consumes.  Per object,
    in the reference's order: ``obj_pose_kitti2nusc`` on the KITTI-convention pose (:638), ``roi_process(roi, H, W, roi_margin, sq_pad=True)``
    on the 2D box (:651; margin 15, clipped to the 1242 x 375 image, so truncated cars give non-square crops), the crop and its mask with the
    background whitened (:654-658).  The camera-in-object pose the renderer wants is the inverse of the converted object pose (:751-754);
    after that the render path is the nuScenes one (the ``kitti2nusc=`` flag of the render functions stays False)."""
    out = []
    margin = hpams.get("roi_margin", 15)
    waymo = hpams.get("dataset", {}).get("name") == "waymo"      # the same loop on 1920 x 1280 images (src/optimizer_waymo.py:125-140)
    for i in ids:
        ob = synthetic.synthetic_kitti_object(i, K=synthetic.WAYMO_K, im_w=synthetic.WAYMO_IM_W, im_h=synthetic.WAYMO_IM_H) if waymo \
            else synthetic.synthetic_kitti_object(i)
        pose_nusc = U.obj_pose_kitti2nusc(ob["obj_pose"][None].clone(), torch.tensor([float(ob["wlh"][2])]))[0]
        R_c2o = pose_nusc[:, :3].T
        cam_pose = torch.cat([R_c2o, -R_c2o @ pose_nusc[:, 3:]], -1)
        roi = U.roi_process(ob["box2d"], ob["im_h"], ob["im_w"], margin, sq_pad=True)
        h, w = int(roi[3] - roi[1]), int(roi[2] - roi[0])
        img, mask = synthetic.synthetic_crop_targets(i, h, w)
        img = img * (mask > 0) + (mask <= 0)
        out.append(dict(wlh=ob["wlh"], obj_diag=np.linalg.norm(ob["wlh"]).astype(np.float32), cam_pose=cam_pose, obj_pose=pose_nusc,
                        K=ob["K"], roi=roi, img=img, mask=mask, index=i))
    return out


def optimize_objects(model, device, n_objects: int, hpams: Optional[dict] = None, rank: int = 0, world_size: int = 1, seed: int = 0,
                     group=None, batch: int = 64, dataset: str = "nusc", return_counts: bool = False, lidar: bool = False):
    """Shard ``n_objects`` synthetic objects over the ranks, optimise the local slice ``batch`` objects per launch
    (``batch=1``: the reference's one-object-at-a-time loop with its global random streams), all-gather the metric rows.
    Returns (n_objects, num_opts*4) on every rank; with ``return_counts`` a last column holds every object's depth-pixel count (the
    ``lidar_pts_cnt`` of the saved results).  ``lidar``: the synthetic objects carry lidar returns with measured depths, so the depth
    column is the reference's depth L1 (src/optimizer_nuscenes.py:751-765)."""
    hpams = hpams or load_hpams(dataset=dataset)
    mine = list(shard_slice(n_objects, world_size, rank))
    objs = make_kitti_objects(mine, hpams) if dataset in ("kitti", "waymo") else make_objects(mine, hpams["render_im_sz"], lidar=lidar)
    rows, counts = [], []

    def start_codes(index):
        g = torch.Generator().manual_seed(seed * 7919 + index)
        return torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3
    if batch <= 1 or hpams.get("sym_aug", 0) or not U._is_native(model):
        for ob in objs:
            sc, tc = start_codes(ob["index"])
            info = {}
            m, *_ = optimize_object(model, device, ob, hpams, sc, tc, seed=seed * 7919 + ob["index"], info=info)
            rows.append(m.reshape(-1)); counts += info["lidar_count"].tolist()
    else:
        for i in range(0, len(objs), batch):
            part = objs[i:i + batch]
            codes = [start_codes(ob["index"]) for ob in part]
            info = {}
            m, *_ = optimize_objects_batched(model, device, part, hpams, torch.cat([c[0] for c in codes]), torch.cat([c[1] for c in codes]),
                                             [seed * 7919 + ob["index"] for ob in part], info=info)
            rows += list(m.reshape(len(part), -1).cpu()); counts += info["lidar_count"].tolist()
    n_cols = hpams["optimize"]["num_opts"] * 4 + (1 if return_counts else 0)
    dev = torch.device(device)
    if return_counts:
        rows = [torch.cat([r, torch.tensor([float(c)])]) for r, c in zip(rows, counts)]
    local = torch.stack(rows).to(dev) if rows else torch.zeros(0, n_cols, device=dev)
    return gather_metric_rows(local, torch.tensor(mine, device=dev, dtype=torch.float32), n_objects, group)
