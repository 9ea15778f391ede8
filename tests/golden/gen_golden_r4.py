#!/usr/bin/env python3
"""Round-4 fixture taken from the reference (build container only: needs /root/reference):

* ``pose_table.npz`` -- the feed-forward pose refinement that fills the optimise loop's pose table
  (``OptimizerNuScenes.fw_pose_update`` / ``fw_pose_one_step``, src/optimizer_nuscenes.py:451-551).  That method lives in a module
  that cannot be imported here (pytorch3d, imageio, skimage, nuscenes-devkit are absent: an ordinary ModuleNotFoundError), so -- as
  ``gen_golden.py`` does for ``vis_scene`` -- this script makes the SAME sequence of calls into the reference's own building blocks:
  ``utils.corners_of_box_batch``, ``utils.view_points_batch``, ``utils.normalize_by_roi`` and the reference's own
  ``SUPNeRF.pose_update`` (model_supnerf, imported with the inert torchvision stand-ins of SURVEY 8c), with the two rotation
  conversions (pytorch3d in the reference, parity unpinned) supplied by the oracle.  It asserts that the oracle's restatement
  reproduces every pose of the table bit for bit and stores inputs + the table.

The pose head's weights are NOT stored (0.5 M floats): both sides make them from the closed formula ``pose_head_formula_params``
below (repeated in the tests).  Data only: numbers, no reference source text.

Usage:  python tests/golden/gen_golden_r4.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import supnerf_oracle as O  # noqa: E402


def pose_head_formula_params(pose_blocks=3, regress_blocks=3, W=256, pose_dim=16):
    """Deterministic pose-head weights from a closed formula (no RNG, no file): w[i, j] = a * sin(0.37 i + 0.11 j + layer), b[i] = 0.01 cos(i)."""
    shapes = [(f"pose_layer_{j}.0", W, pose_dim if j == 0 else W) for j in range(pose_blocks)]
    shapes += [(f"regress_layer_{j}.0", W, 2 * W if j == 0 else W) for j in range(regress_blocks)]
    shapes += [("out_delta_layer", 6, W)]
    out = {}
    for li, (name, n_out, n_in) in enumerate(shapes):
        i = torch.arange(n_out, dtype=torch.float64)[:, None]
        j = torch.arange(n_in, dtype=torch.float64)[None, :]
        scale = (0.02 if name == "out_delta_layer" else 1.0) / np.sqrt(n_in)
        out[name + ".weight"] = (scale * torch.sin(0.37 * i + 0.11 * j + li)).float()
        out[name + ".bias"] = (0.01 * torch.cos(i[:, 0] + li)).float()
    return out


def import_reference():
    cv2 = types.ModuleType("cv2")
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvm = types.ModuleType("torchvision.models"); tvr = types.ModuleType("torchvision.models.resnet")

    class Resize:
        def __init__(self, size):
            self.size = size

        def __call__(self, x):
            return F.interpolate(x, size=self.size, mode="bilinear", align_corners=False)

    class BasicBlock(nn.Module):          # never run here (only the pose head is): the encoder just has to construct
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)

    tvt.Resize = Resize
    tvr.BasicBlock, tvr.Bottleneck = BasicBlock, BasicBlock
    tvr.conv1x1 = lambda i, o, stride=1: nn.Conv2d(i, o, 1, stride, bias=False)
    tvr.conv3x3 = lambda i, o, stride=1, groups=1, dilation=1: nn.Conv2d(i, o, 3, stride, dilation, bias=False)
    tv.transforms, tv.models, tvm.resnet = tvt, tvm, tvr
    for k, v in {"cv2": cv2, "torchvision": tv, "torchvision.transforms": tvt, "torchvision.models": tvm, "torchvision.models.resnet": tvr}.items():
        sys.modules.setdefault(k, v)
    sys.path.insert(0, os.path.join(REF, "src"))
    import model_supnerf  # noqa
    import utils as ref_utils  # noqa
    return model_supnerf, ref_utils


def main():
    MS, RU = import_reference()
    torch.manual_seed(0)
    model = MS.SUPNeRF(shape_blocks=3, texture_blocks=1, pose_blocks=3, regress_blocks=3, latent_dim=256, pose_dim=16)
    head = pose_head_formula_params()
    missing, unexpected = model.load_state_dict(head, strict=False)
    assert not unexpected and all(not k.startswith(("pose_layer", "regress_layer", "out_delta")) for k in missing)
    model.eval()

    B, iters = 3, 3
    g = torch.Generator().manual_seed(4)
    im_feat = torch.randn(B, 256, generator=g) * 0.5
    K = torch.tensor([[1266.4, 0, 816.3], [0, 1266.4, 491.5], [0, 0, 1]]).repeat(B, 1, 1)
    K_inv = torch.linalg.inv(K)
    wlh = torch.tensor([[1.9, 4.6, 1.7], [2.0, 4.9, 1.6], [1.8, 4.3, 1.75]])
    yaw = torch.tensor([0.4, -1.1, 2.3])
    # object poses in the camera frame (nuScenes camera: x right, y down, z forward; object x forward, y left, z up)
    base = torch.tensor([[0., -1., 0.], [0., 0., -1.], [1., 0., 0.]])
    Rz = torch.stack([torch.tensor([[np.cos(a), -np.sin(a), 0.], [np.sin(a), np.cos(a), 0.], [0., 0., 1.]], dtype=torch.float32) for a in yaw])
    R = base[None] @ Rz
    t = torch.tensor([[1.5, 0.8, 14.0], [-4.0, 1.1, 22.0], [6.5, 0.6, 31.0]])[:, :, None]
    src = torch.cat([R, t], dim=2)
    uv = RU.view_points_batch(RU.corners_of_box_batch(src, wlh), K, normalize=True)
    roi = torch.stack([uv[:, 0].min(1).values - 5, uv[:, 1].min(1).values - 5, uv[:, 0].max(1).values + 5, uv[:, 1].max(1).values + 5], 1).round()

    # the reference's sequence (src/optimizer_nuscenes.py:499-507,509-551) on its own building blocks
    table = [src]
    with torch.no_grad():
        for _ in range(iters):
            s = table[-1]
            src_uv = RU.view_points_batch(RU.corners_of_box_batch(s, wlh), K, normalize=True)
            uv_n, dim = RU.normalize_by_roi(src_uv[:, :2, :], roi, need_square=True)
            d = model.pose_update(im_feat, uv_n.view((B, -1)))
            d[:, :3] *= (torch.pi * 2)
            d[:, 3:5] *= dim.unsqueeze(-1)
            d[:, 5] += 1
            pred_R = O.rotvec_to_matrix(O.matrix_to_rotvec(s[:, :, :3]) + d[:, :3])
            T_src = s[:, :, 3:]
            c = torch.matmul(K, T_src)
            pu = c[:, 0] / c[:, 2] + d[:, 3:4]
            pv = c[:, 1] / c[:, 2] + d[:, 4:5]
            pZ = s[:, 2, 3:] * d[:, 5:]
            pT = torch.matmul(K_inv, torch.cat([pu * pZ, pv * pZ, pZ], dim=1).unsqueeze(-1))
            table.append(torch.cat([pred_R, pT], dim=2))
    table = torch.stack(table, dim=1)

    # the oracle's restatement, with ITS pose head from the same state-dict
    tab_o = O.pose_refine_table(lambda f, u: O.pose_head(head, f, u), im_feat, src, wlh, roi, K, K_inv, iters=iters)
    assert torch.equal(tab_o, table), float((tab_o - table).abs().max())
    # and the head alone against the reference's module
    u16 = torch.randn(B, 16, generator=g)
    assert torch.equal(O.pose_head(head, im_feat, u16), model.pose_update(im_feat, u16))
    # the table must actually move (a non-constant pose table is the point of the fixture)
    step = (table[:, 1:] - table[:, :-1]).abs().amax(dim=(2, 3))
    assert float(step.min()) > 1e-3, step
    np.savez_compressed(os.path.join(HERE, "pose_table.npz"), im_feat=im_feat.numpy(), src_pose=src.numpy(), wlh=wlh.numpy(), roi=roi.numpy(),
                        K=K.numpy(), K_inv=K_inv.numpy(), table=table.numpy(), head_uv=u16.numpy(),
                        head_out=model.pose_update(im_feat, u16).detach().numpy(), iters=np.int64(iters))
    print("wrote pose_table.npz; per-step pose change (max abs entry):", step.numpy().round(4).tolist())


if __name__ == "__main__":
    main()
