"""Test helper: the reference's optimise iteration (src/optimizer_nuscenes.py:674-783 == src/optimizer_kitti.py:731-866) written on the
ORACLE renderer, on the CPU, in float32 (the reference's arithmetic) or float64 (the same computation without fp32 rounding: what the
derived tolerance bands of tests/test_full_size.py are measured against).  Used by the trace tests and by
tests/golden/gen_trace_bands.py."""
import numpy as np
import torch

from oracle import supnerf_oracle as O


def oracle_loop(params, obj, hpams, sc0, tc0, seed, reg_iters, pose_noise, D, jitter, dtype=torch.float32):
    """``jitter`` (num_opts, 2, S): the two depth draws of every iteration.  ``D`` = supnerf_amd.driver (host-side rotation helpers and
    the optimiser construction, plain torch).  Returns (num_opts, 3) = [PSNR, rotation error, translation error] per iteration."""
    c = lambda t: t.to(dtype)
    opt = hpams["optimize"]
    rs = np.random.RandomState(seed)
    params = {k: c(v) for k, v in params.items()}
    cam_pose, img, mask, K, jitter = c(obj["cam_pose"]), c(obj["img"]), c(obj["mask"]), c(obj["K"]), c(jitter)
    R_gt = cam_pose[:, :3].T
    t_gt = -R_gt @ cam_pose[:, 3:]
    # the start pose is drawn in float32 whatever the arithmetic of the loop (same start for both)
    R32 = obj["cam_pose"][:, :3].T
    t32 = -R32 @ obj["cam_pose"][:, 3:]
    rot_vec = c(D.matrix_to_axis_angle(R32[None]) + torch.from_numpy(rs.randn(1, 3).astype(np.float32)) * pose_noise[0]).requires_grad_()
    trans_vec = c(t32.T + torch.from_numpy(rs.randn(1, 3).astype(np.float32)) * pose_noise[1]).requires_grad_()
    sc, tc = c(sc0).clone().requires_grad_(), c(tc0).clone().requires_grad_()
    optim = D.make_optimizer(sc, tc, rot_vec, trans_vec, {k: opt[k] for k in ("lr_shape", "lr_texture", "lr_pose")})
    rows = []
    for it in range(opt["num_opts"]):
        optim.zero_grad()
        R = D.axis_angle_to_matrix(rot_vec[0]); t = trans_vec[0].unsqueeze(-1)
        Rc = R.transpose(-2, -1)
        cam2opt = torch.cat([Rc, -Rc @ t], -1)
        out = O.render_rays_v2(params, img, mask, cam2opt, obj["obj_diag"], K, obj["roi"], hpams["n_samples"], sc, tc,
                               True, im_sz=hpams["render_im_sz"], jitter=jitter[it, 0])
        loss, _, _, ps = O.optimise_losses(out[0], out[2], out[3], out[4], hpams["loss_occ_coef"])
        loss.backward()
        pred_R = cam2opt[:, :3].detach().T
        pred_t = -pred_R @ cam2opt[:, 3:].detach()
        rows.append([float(ps), float(D.rot_dist(pred_R, R_gt)), float((pred_t - t_gt).norm())])
        if it > reg_iters:
            optim.step()
    return np.array(rows)
