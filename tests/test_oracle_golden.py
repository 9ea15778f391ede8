"""The CPU oracle against the vectors the REFERENCE produced (tests/golden/*.npz,
made by tests/golden/gen_golden.py in the build container).  Runs anywhere."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O


def same(a, b, tol=0.0):
    a, b = torch.as_tensor(a).detach().double(), torch.as_tensor(b).detach().double()
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= tol if a.numel() else True


def test_weights_are_reproducible(golden, oracle_params):
    h = hashlib.sha256()
    for k in sorted(oracle_params):
        h.update(k.encode())
        h.update(oracle_params[k].numpy().astype(np.float32).tobytes())
    assert h.hexdigest() == str(golden("decoder_b1_s32")["weights_sha256"])
    assert list(oracle_params.keys()) == list(O.decoder_param_names())


def test_positional_encoding_order(golden):
    g = golden("pe")
    same(O.positional_encoding(g["x"], 10), g["pe10"])
    same(O.positional_encoding(g["x"], 4), g["pe4"])
    assert g["pe10"].shape[-1] == 63 and g["pe4"].shape[-1] == 27


@pytest.mark.parametrize("tag", ["b1_s32", "b3_s64", "b2_s7"])
def test_decoder(golden, oracle_params, tag):
    g = golden("decoder_" + tag)
    with torch.no_grad():
        sig, rgb = O.decoder_forward(oracle_params, g["xyz"], g["viewdir"], g["shapecode"], g["texturecode"])
    same(sig, g["sigmas"], 1e-6)
    same(rgb, g["rgbs"], 1e-6)


def test_composite_family(golden):
    g = golden("composite")
    for a, k in zip(O.volume_rendering2(g["sigmas"], g["rgbs"], g["z_shared"]), ("vr2_rgb", "vr2_depth", "vr2_acc")):
        same(a, g[k])
    for a, k in zip(O.composite(g["sigmas"].squeeze(-1), g["rgbs"], g["z_ray"], white_bkgd=True),
                    ("white_rgb", "white_depth", "white_acc")):
        same(a, g[k])
    for a, k in zip(O.volume_rendering3(g["sigmas"], g["rgbs"], g["z_ray"]), ("vr3_rgb", "vr3_depth", "vr3_acc")):
        same(a, g[k])
    B, S = g["z_obj"].shape
    out = O.volume_rendering_batch(g["sigmas"].view(B, -1, S, 1), g["rgbs"].view(B, -1, S, 3), g["z_obj"])
    for a, k in zip(out, ("batch_rgb", "batch_depth", "batch_acc")):
        same(a, g[k])
    # acc_trans excludes the last (1e10 wide) sample; an empty ray keeps it at ~1
    assert abs(float(g["vr2_acc"][0]) - 1.0) < 1e-5
    assert float(g["vr2_acc"][1]) < 1e-6


def test_composite_gradients(golden):
    g = golden("composite_grad")
    sig, rgb, z = [g[k].clone().requires_grad_() for k in ("sigmas", "rgbs", "z")]
    r = O.volume_rendering3(sig, rgb, z, white_bkgd=True)
    ((r[0] * g["w_rgb"]).sum() + (r[1] * g["w_depth"]).sum() + (r[2] * g["w_acc"]).sum()).backward()
    same(sig.grad, g["d_sigmas"], 1e-6)
    same(rgb.grad, g["d_rgbs"], 1e-6)
    same(z.grad, g["d_z"], 1e-5)


def test_rays(golden):
    g = golden("rays")
    o, d = O.pixel_rays(g["K"], g["cam_pose"], g["roi"], uv_steps=[8, 8])
    same(o, g["rays_o"]); same(d, g["viewdir"])
    o, d = O.pixel_rays(g["K"], g["cam_pose"], g["roi_small"])
    same(o, g["rays_o_small"]); same(d, g["viewdir_small"])
    o, d = O.pixel_rays_at(g["K"], g["cam_pose"], g["x_vec"].numpy() + int(g["roi"][0]), g["y_vec"].numpy() + int(g["roi"][1]))
    same(d, g["viewdir_spec"])


@pytest.mark.parametrize("tag", ["a_nusc", "a_demo", "a_kitti"])
def test_family_a_end_to_end(golden, oracle_params, tag):
    g = golden("render_" + tag)
    with torch.no_grad():
        out = O.render_rays_v2(oracle_params, g["img"], g["mask_occ"], g["cam_pose"], np.float32(g["obj_diag"]), g["K"],
                               g["roi"], int(g["n_samples"]), g["shapecode"], g["texturecode"],
                               bool(g["shapenet_obj_cood"]), kitti2nusc=bool(g["kitti2nusc"]),
                               im_sz=int(g["im_sz"]), jitter=g["jitter"])
    for a, k in zip(out, ("rgb", "depth", "acc", "rgb_tgt", "occ")):
        same(a, g[k], 2e-6)


def test_family_a_flip_and_subset(golden, oracle_params):
    g = golden("render_a_flip_subset")
    with torch.no_grad():
        out = O.render_rays_v2(oracle_params, g["img"], g["mask_occ"], g["cam_pose"], np.float32(g["obj_diag"]), g["K"],
                               g["roi"], 64, g["shapecode"], g["texturecode"], True, sym_flip=True, im_sz=8,
                               ray_ids=g["ray_ids"].numpy(), jitter=g["jitter"])
    for a, k in zip(out, ("rgb", "depth", "acc", "rgb_tgt", "occ")):
        same(a, g[k], 2e-6)


def test_resize_targets(golden):
    g = golden("resize_targets")
    im, mk = O.resize_targets(g["img"], g["mask_occ"], 8)
    same(im.reshape(-1, 3), g["rgb_tgt"]); same(mk.reshape(-1, 1), g["occ"])
    assert set(np.unique(mk.numpy()).tolist()) <= {-1.0, 0.0, 1.0}


def test_family_a_specified(golden, oracle_params):
    g = golden("render_a_specified")
    with torch.no_grad():
        out = O.render_rays_specified(oracle_params, g["img"], g["mask_occ"], g["cam_pose"], np.float32(g["obj_diag"]),
                                      g["K"], g["roi"], g["x_vec"].numpy(), g["y_vec"].numpy(), 64, g["shapecode"],
                                      g["texturecode"], True, jitter=g["jitter"])
    for a, k in zip(out, ("rgb", "depth", "acc", "rgb_tgt", "occ")):
        same(a, g[k], 2e-6)


def test_prepare_pixel_samples(golden):
    g = golden("prepare_pixel_samples")
    out = O.prepare_pixel_samples(g["img"], g["mask_occ"], g["cam_pose"], np.float32(g["obj_diag"]), g["K"], g["roi"], 20,
                                  64, True, im_sz=8, ray_ids=g["ray_ids"].numpy(), jitter=g["jitter"])
    for a, k in zip(out, ("xyz", "viewdir", "z_vals", "rgb_tgt", "occ")):
        same(a, g[k])


def test_render_full_img(golden, oracle_params):
    g = golden("render_full_img")
    with torch.no_grad():
        img, depth = O.render_full_img(oracle_params, g["cam_pose"], g["wlh"].numpy(), g["K"], g["roi"], 64, g["shapecode"],
                                       g["texturecode"], True, out_depth=True, jitter=g["jitter"])
    same(img, g["img"], 2e-6); same(depth, g["depth"], 2e-5)


@pytest.mark.parametrize("tag", ["b_hit", "b_s32"])
def test_family_b_end_to_end(golden, oracle_params, tag):
    g = golden("render_" + tag)
    with torch.no_grad():
        out = O.nerf_renderer_render_rays(oracle_params, g["img"], g["mask_occ"], g["cam_pose"], g["wlh"].numpy(), g["K"],
                                          g["roi"], g["shapecode"], g["texturecode"], n_samples=int(g["n_samples"]),
                                          white_bkgd=True, im_sz=int(g["im_sz"]), jitter=g["jitter"])
        ro, vd = O.pixel_rays(g["K"], g["cam_pose"], g["roi"], uv_steps=[int(g["im_sz"])] * 2)
        _, _, zv, hit = O.aabb_sampled_rays(ro, vd, g["wlh"].numpy(), int(g["n_samples"]), g["jitter"])
    for a, k in zip(out, ("rgb", "depth", "acc", "rgb_tgt", "occ")):
        same(a, g[k], 2e-6)
    assert torch.equal(hit, g["hit"].bool())
    same(zv, g["z_vals"], 1e-6)
    # rays that miss: all samples collapse to z = diag/2 and only the white background shows
    miss = ~hit
    assert miss.any() and hit.any()


def test_render_rays_v3(golden, oracle_params):
    g, g3 = golden("render_b_hit"), golden("render_v3_b_hit")
    with torch.no_grad():
        out = O.render_rays_v3(oracle_params, g["img"], g["mask_occ"], g["cam_pose"], g["wlh"].numpy(), g["K"], g["roi"],
                               64, g["shapecode"], g["texturecode"], True, im_sz=8, adjust_scale=float(g3["adjust_scale"]),
                               jitter=g3["jitter"])
    for a, k in zip(out[:3], ("rgb", "depth", "acc")):
        same(a, g3[k], 3e-5)
    with pytest.raises(ValueError):
        O.render_rays_v3(oracle_params, g["img"], g["mask_occ"], g["cam_pose"], g["wlh"].numpy(), g["K"], g["roi"], 32,
                         g["shapecode"], g["texturecode"], True, im_sz=8)


def test_gradients_family_a(golden, oracle_params):
    g = golden("grads_family_a")
    sc, tc, pose = [g[k].clone().requires_grad_() for k in ("shapecode", "texturecode", "cam_pose")]
    p = {k: v.clone().requires_grad_() for k, v in oracle_params.items()}
    out = O.render_rays_v2(p, g["img"], g["mask_occ"], pose, np.float32(g["obj_diag"]), g["K"], g["roi"], 64, sc, tc, True,
                           im_sz=8, jitter=g["jitter"])
    loss, l_rgb, l_occ, psnr = O.optimise_losses(out[0], out[2], g["img"].reshape(-1, 3), g["mask_occ"].reshape(-1, 1), 0.1)
    loss.backward()
    same(loss, g["loss"], 1e-6); same(psnr, g["psnr"], 1e-4)
    same(sc.grad, g["d_shapecode"], 1e-7); same(tc.grad, g["d_texturecode"], 1e-7); same(pose.grad, g["d_cam_pose"], 1e-6)
    for k, v in p.items():
        key = k.replace(".", "_")
        s = g["dWsum_" + key]
        assert abs(float(v.grad.double().sum()) - float(s[0])) <= 1e-5 * max(1.0, float(s[1]))
        if v.dim() == 2:
            same(v.grad[0], g["dWrow0_" + key], 1e-6)


def test_gradients_family_b(golden, oracle_params):
    g = golden("grads_family_b")
    sc, tc, pose = [g[k].clone().requires_grad_() for k in ("shapecode", "texturecode", "cam_pose")]
    out = O.nerf_renderer_render_rays(oracle_params, g["img"], g["mask_occ"], pose, g["wlh"].numpy(), g["K"], g["roi"], sc, tc,
                                      n_samples=32, white_bkgd=True, im_sz=8, jitter=g["jitter"])
    loss = O.optimise_losses(out[0], out[2], out[3], out[4], 0.1)[0] + 0.01 * out[1].sum()
    loss.backward()
    same(loss, g["loss"], 1e-6)
    same(sc.grad, g["d_shapecode"], 1e-7); same(pose.grad, g["d_cam_pose"], 1e-6)


def test_training_shape_step(golden, oracle_params):
    g = golden("train_step")
    B, n, S = g["xyz"].shape[:3]
    sc, tc = g["shapecode"].clone().requires_grad_(), g["texturecode"].clone().requires_grad_()
    p = {k: v.clone().requires_grad_() for k, v in oracle_params.items()}
    sig, rgb = O.decoder_forward(p, g["xyz"].flatten(0, 1), g["viewdir"].flatten(0, 1), sc, tc)
    out = O.volume_rendering_batch(sig.view(B, n, S, 1), rgb.view(B, n, S, 3), g["z_vals"])
    loss = ((out[0] - g["tgt"]) ** 2).mean() + 0.1 * out[2].mean()
    loss.backward()
    same(loss, g["loss"], 1e-6)
    same(sc.grad, g["d_shapecode"], 1e-7)
    for k, v in p.items():
        key = k.replace(".", "_")
        if v.dim() == 2:
            same(v.grad[0], g["dWrow0_" + key], 1e-6)


def test_scene_compositing(golden, oracle_params):
    """Multi-object scene (scripts/demo.py:425-579): ray table, per-pixel depth merge + white-background composite, canvas."""
    g = golden("scene")
    H, W, S, bs = int(g["H"]), int(g["W"]), int(g["n_samples"]), int(g["ray_batch_size"])
    tab, valid, diags = O.scene_rays(g["obj_poses"], g["obj_wlh"], g["K"], H, W)
    assert torch.equal(valid, g["valid"].bool())
    out = O.scene_composite(g["b0_sigmas"], g["b0_rgbs"], g["b0_z"])
    same(out[0], g["b0_rgb"]); same(out[1], g["b0_depth"]); same(out[2], g["b0_acc"])
    jit = list(torch.split(g["jitter"], [int(r) for r in g["jitter_rows"]]))
    canvas, img = O.vis_scene(oracle_params, g["obj_poses"], g["obj_wlh"], g["shapecodes"], g["texturecodes"], g["K"], H, W, S,
                              ray_batch_size=bs, jitters=jit)
    same(canvas, g["canvas"])
    assert np.array_equal(img, g["image"].numpy())
    # a proper stable merge gives the same picture as the reference's searchsorted scatter (equal depths only occur on empty samples)
    z = g["b0_z"]
    order = torch.argsort(z, dim=1, stable=True)
    ref2 = O.composite(torch.gather(g["b0_sigmas"], 1, order), torch.gather(g["b0_rgbs"], 1, order[:, :, None].repeat(1, 1, 3)),
                       torch.gather(z, 1, order), white_bkgd=True)
    same(ref2[0], g["b0_rgb"], 1e-6)


def test_render_api_twins(golden, oracle_params):
    """The rest of the render API (src/renderer.py:169-352, src/utils.py:94-104,154-199,236-280) against the reference's outputs."""
    g = golden("twins")
    img, mask, pose, wlh, K, roi, sc, tc = [g[k] for k in ("img", "mask_occ", "cam_pose", "wlh", "K", "roi", "shapecode", "texturecode")]
    wlh = wlh.numpy()
    with torch.no_grad():
        out = O.nerf_renderer_render_rays_specified(oracle_params, img, mask, pose, wlh, K, roi, g["spec_x"].numpy(), g["spec_y"].numpy(), sc, tc,
                                                    n_samples=32, jitter=g["spec_jitter"])
        for a, k in zip(out, ("spec_rgb", "spec_depth", "spec_acc", "spec_tgt", "spec_occ")):
            same(a, g[k])
        out = O.nerf_renderer_prepare_pixel_samples(img, mask, pose, wlh, K, roi, 40, n_samples=32, im_sz=8, ray_ids=g["pps_ids"].numpy(), jitter=g["pps_jitter"])
        for a, k in zip(out, ("pps_xyz", "pps_viewdir", "pps_z", "pps_tgt", "pps_occ")):
            same(a, g[k])
        full = O.nerf_renderer_render_full_img(oracle_params, pose, wlh, K, g["full_roi"], sc, tc, n_samples=32, out_depth=True, jitter=g["full_jitter"])
        same(full[0], g["full_img"]); same(full[1], g["full_depth"])
        views = O.nerf_renderer_render_virtual_imgs(oracle_params, wlh, K, sc, tc, n_samples=32, radius=12., pan_num=2, img_sz=12, jitters=list(g["virt_b_jitter"]))
        same(torch.stack(views), g["virt_b"])
    same(O.shared_depth_samples(7.5, 12.25, 9, g["util_jitter"]), g["util_z"])
    same(torch.linspace(7.5, 12.25, 9), g["util_z_fixed"])                   # z_fixed: plain linspace(near, far)
    leg = O.volume_rendering_legacy(g["legacy_sig"], g["legacy_rgb"], g["util_z"])
    same(leg[0], g["legacy_out_rgb"]); same(leg[1], g["legacy_out_depth"])
    so, sd = O.srn_rays(6, 5, 40.0, g["srn_c2w"])
    same(so, g["srn_o"]); same(sd, g["srn_d"])
    tn, tf, hit = O.slab_intersect(g["box_o"], g["box_d"], -g["box_max"], g["box_max"])
    assert torch.equal(hit, g["box_hit"].bool()) and torch.equal(tn[hit], g["box_z_in"]) and torch.equal(tf[hit], g["box_z_out"])
    same(O.unit_interval_samples(g["sfr_rays"][:, 6:7], g["sfr_rays"][:, 7:8], 32, g["sfr_jitter"]), g["sfr_z"])
    psr = O.aabb_sampled_rays(g["util_rays_o"], g["util_rays_d"], wlh, 32, g["psr_jitter"])
    same(psr[0], g["psr_xyz"]); same(psr[2], g["psr_z"]); assert torch.equal(psr[3], g["psr_hit"].bool())
    vr = O.composite(g["vr_sig"], g["vr_rgb"], g["psr_z"], white_bkgd=True)
    same(vr[0], g["vr_out_rgb"]); same(vr[1], g["vr_out_depth"]); same(vr[2], g["vr_out_acc"])
    vb = O.composite(g["vrb_sig"], g["vrb_rgb"], g["vrb_z"], white_bkgd=False)
    same(vb[0], g["vrb_out_rgb"]); same(vb[1], g["vrb_out_depth"]); same(vb[2], g["vrb_out_acc"])


# ------------------------------------------------------------------ round-2 fixtures (tests/golden/gen_golden_r2.py)
def test_kitti_pose_convention_and_roi(golden):
    g = golden("kitti")
    assert torch.equal(O.obj_pose_kitti2nusc(g["k2n_in"], g["k2n_h"]), g["k2n_out"])
    for b, H, W, m, sq, want in zip(g["roi_in"], g["roi_H"], g["roi_W"], g["roi_margin"], g["roi_sq"], g["roi_out"]):
        got = O.roi_process(b, None if H < 0 else int(H), None if W < 0 else int(W), int(m), bool(sq))
        assert torch.equal(got, want), (b.tolist(), int(H), int(W), int(m), int(sq), got.tolist(), want.tolist())
    assert torch.equal(O.unit_interval_samples(g["sfr2_rays"][:, 6:7], g["sfr2_rays"][:, 7:8], 16, g["sfr2_jitter"]), g["sfr2_z"])


def test_kitti_object_end_to_end(golden, oracle_params):
    """render_rays_v2 on a truncated KITTI car whose crop is not im_sz^2: bilinear resize + int32 mask truncation + render."""
    g = golden("kitti")
    import supnerf_amd as A                       # host-side object generator only (no compute)
    ob = A.driver.make_kitti_objects([int(g["e2e_index"])], A.driver.load_hpams(dataset="kitti"))[0]
    assert torch.equal(ob["roi"], g["e2e_roi"]) and torch.equal(ob["cam_pose"], g["e2e_cam_pose"])
    with torch.no_grad():
        out = O.render_rays_v2(oracle_params, ob["img"], ob["mask"], ob["cam_pose"], ob["obj_diag"], ob["K"], ob["roi"], 64, g["e2e_shapecode"],
                               g["e2e_texturecode"], True, im_sz=16, jitter=g["e2e_jitter"])
    for a, k in zip(out, ("e2e_rgb", "e2e_depth", "e2e_acc", "e2e_tgt", "e2e_occ")):
        assert torch.equal(a, g[k]), k


def test_eval_reader_restatement(golden):
    """O.eval_curves == what the reference's collect_eval_results plotted from a file written by supnerf_amd.io."""
    g = golden("formats")
    rows = g["eval_rows"]
    saved = {"psnr_eval": {}, "depth_err_mean": {}, "lidar_pts_cnt": {}, "R_eval": {}, "T_eval": {}}
    for r, i in zip(rows, g["eval_ids"].tolist()):
        k = f"{i}_0"
        saved["psnr_eval"][k], saved["depth_err_mean"][k] = r[:, 0].tolist(), r[:, 1].tolist()
        saved["R_eval"][k], saved["T_eval"][k], saved["lidar_pts_cnt"][k] = list(r[:, 2].unbind(0)), list(r[:, 3].unbind(0)), 64
    for a, k in zip(O.eval_curves(saved, rows.shape[1]), ("eval_psnr", "eval_depth", "eval_R_deg", "eval_T")):
        assert np.allclose(a, g[k].numpy(), rtol=0, atol=1e-12), k

