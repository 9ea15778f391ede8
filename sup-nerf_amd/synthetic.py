"""Synthetic "nuScenes car" workloads: seeded decoder weights, car-sized boxes at 8-35 m with
a plausible camera, square rois, random targets with an elliptical occupancy mask.  No dataset or checkpoint is
available offline, so bench.py, the driver and the tests all draw from here.  (The CPU oracle keeps an identical,
independent copy so that it does not depend on the product package; tests/test_host_logic.py checks they agree.)"""
import math

import numpy as np
import torch

WLH_MEAN = np.array([1.94, 4.64, 1.71], dtype=np.float32)      # src/optimizer_nuscenes.py:27 of the reference
WLH_STD = np.array([0.19, 0.46, 0.25], dtype=np.float32)
NUSC_K = np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]], dtype=np.float32)
KITTI_K = np.array([[721.5, 0.0, 609.6], [0.0, 721.5, 172.9], [0.0, 0.0, 1.0]], dtype=np.float32)
WAYMO_K = np.array([[2055.6, 0.0, 939.7], [0.0, 2055.6, 641.1], [0.0, 0.0, 1.0]], dtype=np.float32)      # a typical FRONT camera, 1920 x 1280
WAYMO_IM_W, WAYMO_IM_H = 1920, 1280


def decoder_layer_shapes(shape_blocks=3, texture_blocks=1, W=256, latent_dim=256, d_xyz=63, d_dir=27):
    """(name, n_out, n_in) in the reference's registration order (src/model_supnerf.py:184-199)."""
    out = [("encoding_xyz.0", W, d_xyz)]
    for j in range(1, shape_blocks + 1):
        out += [(f"shape_latent_layer_{j}.0", W, latent_dim), (f"shape_layer_{j}.0", W, W)]
    out += [("encoding_shape", W, W), ("sigma.0", 1, W), ("encoding_viewdir.0", W, W + d_dir)]
    for j in range(1, texture_blocks + 1):
        out += [(f"texture_latent_layer_{j}.0", W, latent_dim), (f"texture_layer_{j}.0", W, W)]
    out += [("rgb.0", W // 2, W), ("rgb.2", 3, W // 2)]
    return out


def init_decoder_params(shape_blocks=3, texture_blocks=1, seed=0, sigma_bias=-2.0):
    """nn.Linear default init, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias, from one seeded generator;
    the density-head bias is then set to ``sigma_bias`` so that alpha spans (0,1)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, n_out, n_in in decoder_layer_shapes(shape_blocks, texture_blocks):
        bound = 1.0 / math.sqrt(n_in)
        p[name + ".weight"] = (torch.rand(n_out, n_in, generator=g) * 2 - 1) * bound
        p[name + ".bias"] = (torch.rand(n_out, generator=g) * 2 - 1) * bound
    if sigma_bias is not None:
        p["sigma.0.bias"] = torch.full((1,), float(sigma_bias))
    return p


def synthetic_object(index, im_w=1600, im_h=900, K=NUSC_K):
    """Deterministic in ``index``: dict(wlh, obj_diag, cam_pose (3,4) camera-in-object, K, roi int32[4])."""
    rs = np.random.RandomState(1000 + index)
    wlh = (WLH_MEAN + WLH_STD * rs.randn(3)).astype(np.float32)
    diag = np.linalg.norm(wlh).astype(np.float32)
    yaw = rs.uniform(-np.pi, np.pi)
    depth = rs.uniform(8.0, 35.0)
    lateral = rs.uniform(-0.25, 0.25) * depth
    c, s = np.cos(yaw), np.sin(yaw)
    R_obj = np.array([[c, -s, 0], [0, 0, -1], [s, c, 0]], dtype=np.float32)      # object axes in the camera frame
    t_obj = np.array([lateral, 1.2, depth], dtype=np.float32)
    R_c2o = R_obj.T
    t_c2o = -R_c2o @ t_obj
    cam_pose = torch.from_numpy(np.concatenate([R_c2o, t_c2o[:, None]], axis=1).astype(np.float32))
    u = K[0, 0] * t_obj[0] / t_obj[2] + K[0, 2]
    v = K[1, 1] * t_obj[1] / t_obj[2] + K[1, 2]
    half = int(max(16, 0.36 * K[0, 0] * diag / depth))
    x0 = int(np.clip(u - half, 0, im_w - 2 * half - 1))
    y0 = int(np.clip(v - half, 0, im_h - 2 * half - 1))
    roi = torch.tensor([x0, y0, x0 + 2 * half, y0 + 2 * half], dtype=torch.int32)
    return dict(wlh=wlh, obj_diag=diag, cam_pose=cam_pose, K=torch.from_numpy(K.copy()), roi=roi)


def synthetic_targets(index, im_sz):
    """Target crop (im_sz,im_sz,3) in [0,1) and occupancy mask (im_sz,im_sz,1) in {-1,0,1} (ellipse, thin unknown rim)."""
    g = torch.Generator().manual_seed(2000 + index)
    img = torch.rand(im_sz, im_sz, 3, generator=g)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, im_sz), torch.linspace(-1, 1, im_sz), indexing="ij")
    r = (xx / 0.8) ** 2 + (yy / 0.55) ** 2
    mask = torch.where(r < 1.0, torch.ones_like(r), -torch.ones_like(r))
    mask = torch.where((r >= 1.0) & (r < 1.3), torch.zeros_like(r), mask)
    return img, mask[..., None]


# ------------------------------------------------------------------ KITTI-convention objects (BASELINE config 4)
KITTI_IM_W, KITTI_IM_H = 1242, 375


def _kitti_corners(obj_pose, wlh):
    """The 8 box corners in the camera frame for a KITTI-convention pose (x front, y down, z left, origin at the box bottom:
    src/utils.py:1086-1090)."""
    w, l, h = [float(v) for v in wlh]
    x = l / 2 * np.array([1, 1, 1, 1, -1, -1, -1, -1], dtype=np.float64)
    y = h / 2 * np.array([-2, -2, 0, 0, -2, -2, 0, 0], dtype=np.float64)
    z = w / 2 * np.array([1, -1, -1, 1, 1, -1, -1, 1], dtype=np.float64)
    return obj_pose[:, :3].astype(np.float64) @ np.vstack([x, y, z]) + obj_pose[:, 3:].astype(np.float64)


def synthetic_kitti_object(index, K=KITTI_K, im_w=KITTI_IM_W, im_h=KITTI_IM_H):
    """Deterministic in ``index``: a car as a KITTI label would describe it (src/data_kitti.py:430-449): ``obj_pose`` (3,4) =
    [R_y(ry) | t] with t the bottom centre of the box in the camera frame (x right, y down, z forward), ``wlh``, the tight 2D box
    ``box2d`` int32[4] of the projected corners clipped to the 1242 x 375 image, K = a typical P2[:, :3].  The pose is still in the
    KITTI convention: ``utils.obj_pose_kitti2nusc`` converts it on the host like src/optimizer_kitti.py:638-639."""
    rs = np.random.RandomState(5000 + index)
    wlh = (WLH_MEAN + WLH_STD * rs.randn(3)).astype(np.float32)
    ry = rs.uniform(-np.pi, np.pi)
    depth = rs.uniform(8.0, 35.0)
    lateral = rs.uniform(-0.22, 0.22) * depth
    c, s = np.cos(ry), np.sin(ry)
    R_obj = np.array([[c, 0., s], [0., 1., 0.], [-s, 0., c]], dtype=np.float32)
    t_obj = np.array([lateral, 1.65, depth], dtype=np.float32)          # camera 1.65 m above the road
    obj_pose = np.concatenate([R_obj, t_obj[:, None]], axis=1).astype(np.float32)
    uvw = K.astype(np.float64) @ _kitti_corners(obj_pose, wlh)
    u, v = uvw[0] / uvw[2], uvw[1] / uvw[2]
    box = np.array([np.floor(u.min()), np.floor(v.min()), np.ceil(u.max()), np.ceil(v.max())])
    box = np.clip(box, [0, 0, 0, 0], [im_w - 1, im_h - 1, im_w - 1, im_h - 1]).astype(np.int32)
    return dict(wlh=wlh, obj_pose=torch.from_numpy(obj_pose), K=torch.from_numpy(K.copy()), box2d=torch.from_numpy(box),
                im_w=im_w, im_h=im_h)


def synthetic_crop_targets(index, h, w):
    """Target crop (h,w,3) in [0,1) and occupancy mask (h,w,1) in {-1,0,1} of a (possibly non-square) roi: what the KITTI loop cuts
    out of the image and its instance mask (src/optimizer_kitti.py:651-656) before ``render_rays_v2`` resizes it."""
    g = torch.Generator().manual_seed(7000 + index)
    img = torch.rand(h, w, 3, generator=g)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    r = (xx / 0.8) ** 2 + (yy / 0.6) ** 2
    mask = torch.where(r < 1.0, torch.ones_like(r), -torch.ones_like(r))
    mask = torch.where((r >= 1.0) & (r < 1.3), torch.zeros_like(r), mask)
    return img, mask[..., None]


def synthetic_lidar(index, mask, ob):
    """Synthetic lidar returns on the foreground of a crop: ``xy`` (n,2) integer crop pixels (x, y) and ``depth`` (n,) metres, what the
    reference cuts out of its projected depth map (src/optimizer_nuscenes.py:751-754: ``gt_depth_map > 0`` on the mask).  The count differs
    from object to object like real sweeps (40 + 13 (index mod 5), capped by the foreground); depths lie on a smooth bump in front of the
    object's centre plus 2 cm of range noise."""
    rs = np.random.RandomState(9000 + index)
    ys, xs = np.where(mask[:, :, 0].numpy() > 0)
    n = min(40 + 13 * (index % 5), len(ys))
    pick = rs.permutation(len(ys))[:n]
    x, y = xs[pick], ys[pick]
    h, w = mask.shape[0], mask.shape[1]
    r2 = ((x / max(w - 1, 1)) * 2 - 1) ** 2 + ((y / max(h - 1, 1)) * 2 - 1) ** 2
    dist = float(np.linalg.norm(ob["cam_pose"][:, 3].numpy()))
    depth = dist - 0.3 * float(ob["obj_diag"]) * (1.0 - 0.5 * r2) + 0.02 * rs.randn(n)
    return np.stack([x, y], axis=1).astype(np.int64), depth.astype(np.float32)
