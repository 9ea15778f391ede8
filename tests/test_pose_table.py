"""The pose table of the optimise loop's render-only iterations (src/optimizer_nuscenes.py:641-668,684-689) and the feed-forward pose
refinement that fills it (``fw_pose_update`` / ``fw_pose_one_step``, :451-551): fixture ``pose_table.npz`` was made by the reference's own
building blocks + its own ``SUPNeRF.pose_update`` (tests/golden/gen_golden_r4.py)."""
import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O


def pose_head_formula_params(pose_blocks=3, regress_blocks=3, W=256, pose_dim=16):
    """The generator's closed-formula pose-head weights (tests/golden/gen_golden_r4.py): w[i,j] = a sin(0.37 i + 0.11 j + layer)."""
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


def test_oracle_pose_refinement_reproduces_the_references_table(golden):
    g = golden("pose_table")
    head = pose_head_formula_params()
    assert torch.equal(O.pose_head(head, g["im_feat"], g["head_uv"]), g["head_out"])             # SUPNeRF.pose_update, bit for bit
    tab = O.pose_refine_table(lambda f, u: O.pose_head(head, f, u), g["im_feat"], g["src_pose"], g["wlh"], g["roi"], g["K"], g["K_inv"],
                              iters=int(g["iters"]))
    assert torch.equal(tab, g["table"])
    assert float((g["table"][:, 1:] - g["table"][:, :-1]).abs().amax(dim=(2, 3)).min()) > 1e-3    # the table moves


def test_driver_pose_refinement_with_the_stock_pose_head(golden):
    """``driver.fw_pose_update`` drives ``model.pose_update`` -- stock PyTorch Linear layers of the drop-in SUPNeRF -- like the reference
    iterates its head; its own Rodrigues conversions differ from the oracle's in rounding only."""
    import supnerf_amd as A
    g = golden("pose_table")
    model = A.SUPNeRF(shape_blocks=3, texture_blocks=1, pose_blocks=3, regress_blocks=3, img_encoder=False)
    missing, unexpected = model.load_state_dict(pose_head_formula_params(), strict=False)
    assert not unexpected
    assert torch.allclose(model.pose_update(g["im_feat"], g["head_uv"]), g["head_out"], atol=1e-6)
    tab = A.driver.fw_pose_update(model, g["im_feat"], g["src_pose"], g["wlh"], g["roi"], g["K"], g["K_inv"], iters=int(g["iters"]))
    assert tab.shape == g["table"].shape
    assert float((tab[..., :3] - g["table"][..., :3]).abs().max()) < 2e-6 and float((tab[..., 3] - g["table"][..., 3]).abs().max()) < 2e-4


def test_pose_table_shape_is_checked():
    import supnerf_amd as A
    with pytest.raises(A.SnrError, match="pose_per_iter"):
        A.driver.optimize_object(None, "cpu", {}, A.driver.load_hpams(), None, None, reg_iters=3, pose_per_iter=torch.zeros(2, 3, 4))


def _oracle_loop_with_table(params, obj, hpams, sc0, tc0, seed, reg_iters, table, D, jitter):
    """The reference iteration (src/optimizer_nuscenes.py:674-783) on the oracle renderer with the pose table: iterations <= reg_iters
    render at table[it], rot_vec / trans_vec start from table[-1] (:652,664-666,684-689)."""
    opt = hpams["optimize"]
    R_gt = obj["cam_pose"][:, :3].T
    t_gt = -R_gt @ obj["cam_pose"][:, 3:]
    rot_vec = D.matrix_to_axis_angle(table[-1, :3, :3][None]).clone().requires_grad_()
    trans_vec = table[-1, :3, 3][None].clone().requires_grad_()
    sc, tc = sc0.clone().requires_grad_(), tc0.clone().requires_grad_()
    optim = D.make_optimizer(sc, tc, rot_vec, trans_vec, {k: opt[k] for k in ("lr_shape", "lr_texture", "lr_pose")})
    rows = []
    for it in range(opt["num_opts"]):
        optim.zero_grad()
        if it > reg_iters:
            R, t = D.axis_angle_to_matrix(rot_vec[0]), trans_vec[0].unsqueeze(-1)
        else:
            R, t = table[it, :3, :3], table[it, :3, 3:]
        Rc = R.transpose(-2, -1)
        cam2opt = torch.cat([Rc, -Rc @ t], -1)
        out = O.render_rays_v2(params, obj["img"], obj["mask"], cam2opt, obj["obj_diag"], obj["K"], obj["roi"], hpams["n_samples"], sc, tc,
                               True, im_sz=hpams["render_im_sz"], jitter=jitter[it, 0])
        loss, _, _, ps = O.optimise_losses(out[0], out[2], out[3], out[4], hpams["loss_occ_coef"])
        loss.backward()
        pred_R = cam2opt[:, :3].detach().T
        rows.append([float(ps), float(D.rot_dist(pred_R, R_gt)), float((-pred_R @ cam2opt[:, 3:].detach() - t_gt).norm())])
        if it > reg_iters:
            optim.step()
    return np.array(rows)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "auto"])
def test_fused_loop_api_loop_and_oracle_loop_agree_on_a_moving_pose_table(golden, precision):
    """A non-constant pose table (the stock pose head's own per-step poses): fused loop == API-structured loop == oracle loop."""
    import supnerf_amd as A
    D = A.driver
    dev = torch.device("cuda:0")
    params = O.init_decoder_params()
    model = A.SUPNeRF(shape_blocks=3, texture_blocks=1, pose_blocks=3, regress_blocks=3, img_encoder=False)
    model.load_state_dict({**params, **pose_head_formula_params()}, strict=True)
    model = model.to(dev)
    model.precision = precision
    hp = D.load_hpams()
    hp["render_im_sz"] = 16
    hp["optimize"]["num_opts"] = 9
    reg_iters = 3
    obj = D.make_objects([21], 16)[0]
    # the table: the STOCK pose head iterated from a perturbed object pose, on the GPU (src/optimizer_nuscenes.py:641-651)
    R_gt = obj["cam_pose"][:, :3].T
    t_gt = -R_gt @ obj["cam_pose"][:, 3:]
    start = torch.cat([D.axis_angle_to_matrix(D.matrix_to_axis_angle(R_gt[None]) + torch.tensor([[0.04, -0.03, 0.05]]))[0], t_gt + torch.tensor([[0.2], [-0.1], [0.4]])], -1)
    wlh = torch.tensor([[1.9, 4.6, 1.7]])
    K = obj["K"][None].float()
    gen = torch.Generator().manual_seed(8)
    im_feat = torch.randn(1, 256, generator=gen) * 0.5
    roi = torch.as_tensor(np.asarray(obj["roi"]), dtype=torch.float32)[None]
    table = D.fw_pose_update(model, im_feat.to(dev), start[None].to(dev), wlh.to(dev), roi.to(dev), K.to(dev), torch.linalg.inv(K).to(dev),
                             iters=reg_iters)[0].cpu()
    assert table.shape == (reg_iters + 1, 3, 4) and float((table[1:] - table[:-1]).abs().amax(dim=(1, 2)).min()) > 1e-4
    sc0, tc0 = torch.randn(1, 256, generator=gen) * 0.3, torch.randn(1, 256, generator=gen) * 0.3
    jitter = torch.rand(9, 2, 64, generator=gen)
    ref = _oracle_loop_with_table(params, obj, hp, sc0, tc0, 9, reg_iters, table, D, jitter)
    m_f, sc_f, tc_f, pose_f = D.optimize_object(model, dev, obj, hp, sc0, tc0, reg_iters=reg_iters, seed=9, jitter=jitter, pose_per_iter=table)
    m_a, sc_a, tc_a, pose_a = D.optimize_object_api(model, dev, obj, hp, sc0, tc0, reg_iters=reg_iters, seed=9, jitter=jitter, pose_per_iter=table)
    for m in (m_f, m_a):
        got = m[:, [0, 2, 3]].numpy()
        # render-only iterations: the table's poses, identical renders up to kernel round-off; then Adam-amplified fp32 drift
        assert np.abs(got[:reg_iters + 1, 0] - ref[:reg_iters + 1, 0]).max() < 1e-3, (got, ref)
        assert np.abs(got[:reg_iters + 1, 1:] - ref[:reg_iters + 1, 1:]).max() < 1e-5
        assert np.abs(got[:, 0] - ref[:, 0]).max() < 0.05 and np.abs(got[:, 1] - ref[:, 1]).max() < 2e-3 and np.abs(got[:, 2] - ref[:, 2]).max() < 2e-2
    # the rotation / translation errors of the render-only iterations differ from one another: the table is what was rendered
    assert len({round(float(v), 6) for v in m_f[:reg_iters + 1, 2]}) == reg_iters + 1
    assert float((m_f[:, [0, 2, 3]] - m_a[:, [0, 2, 3]]).abs().max()) < 0.05
    assert float((pose_f.cpu() - pose_a.cpu()).abs().max()) < 5e-3
