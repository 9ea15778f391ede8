"""The single-launch pieces of the optimise iteration (loss tail, pose -> rays, metric row, AdamW) against plain PyTorch fp32 references of
the same formulas -- the reference's own lines are cited at each -- and the fused loop against the loop written on the public API."""
import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def amd():
    import supnerf_amd
    return supnerf_amd


def md(a, b):
    return float((a.detach().double().cpu() - torch.as_tensor(b).detach().double().cpu()).abs().max())


def rel(a, b):
    b = torch.as_tensor(b).detach().double().cpu()
    return md(a, b) / (float(b.abs().max()) + 1e-30)


# ------------------------------------------------------------------ loss tail (src/optimizer_nuscenes.py:729-744)
@pytest.mark.parametrize("B,n", [(1, 4096), (3, 1024), (5, 37), (64, 256)])
def test_loss_tail_forward_backward(amd, dev, B, n):
    g = torch.Generator().manual_seed(B * 1000 + n)
    rgb = torch.rand(B * n, 3, generator=g, dtype=torch.float64) * 1.2 - 0.1
    acc = torch.rand(B * n, generator=g, dtype=torch.float64)
    tgt = torch.rand(B * n, 3, generator=g, dtype=torch.float64)
    occ = (torch.randint(0, 3, (B * n, 1), generator=g) - 1).double()
    up = torch.rand(B, generator=g, dtype=torch.float64) + 0.5
    rgb_r, acc_r = rgb.clone().requires_grad_(), acc.clone().requires_grad_()
    rows = []
    for b in range(B):            # the reference's formulas in float64, object by object
        sl = slice(b * n, (b + 1) * n)
        loss, l_rgb, l_occ, psnr = O.optimise_losses(rgb_r[sl], acc_r[sl], tgt[sl], occ[sl], 0.1)
        rows.append(torch.stack([loss, l_rgb, l_occ, 10 ** (-psnr / 10)]))
    want = torch.stack(rows)
    (want[:, 0] * up).sum().backward()
    rgb_d, acc_d = rgb.float().to(dev).requires_grad_(), acc.float().to(dev).requires_grad_()
    loss, metrics = amd.ops.LossTail.apply(rgb_d, acc_d, tgt.float().to(dev), occ.float().to(dev), 0.1, n)
    assert loss.shape == (B,) and metrics.shape == (B, 3) and not metrics.requires_grad
    assert rel(loss, want[:, 0]) < 2e-6 and rel(metrics, want[:, 1:]) < 2e-6
    (loss * up.float().to(dev)).sum().backward()
    assert rel(rgb_d.grad, rgb_r.grad) < 2e-6 and rel(acc_d.grad, acc_r.grad) < 2e-6
    # an object without any labelled pixel: the 1e-9 in the denominators keeps everything finite (0 / 1e-9)
    z = torch.zeros(n, 1, device=dev)
    l0, m0 = amd.ops.LossTail.apply(rgb_d.detach()[:n], acc_d.detach()[:n], tgt.float().to(dev)[:n], z, 0.1, n)
    assert float(l0) == 0.0 and float(m0.abs().max()) == 0.0


def test_loss_tail_rejects_partial_objects(amd, dev):
    with pytest.raises(amd.SnrError):
        amd.ops.loss_tail_fwd(torch.rand(10, 3, device=dev), torch.rand(10, device=dev), torch.rand(10, 3, device=dev), torch.rand(10, 1, device=dev), 0.1, 4)
    with pytest.raises(amd.SnrError):
        amd.ops.loss_tail_fwd(torch.rand(8, 3), torch.rand(8), torch.rand(8, 3), torch.rand(8, 1), 0.1, 4)       # CPU tensors


# ------------------------------------------------------------------ pose -> rays (src/optimizer_nuscenes.py:685-699, src/utils.py:107-135,159-164,468-469)
def torch_pose_rays(D, U, rot_vec, trans_vec, cam, half, jit, S, opt_cam_pose):
    R = D.axis_angle_to_matrix(rot_vec)
    t = trans_vec.unsqueeze(-1)
    if not opt_cam_pose:
        Rc = R.transpose(-2, -1)
        c2o = torch.cat([Rc, -Rc @ t], -1)
    else:
        c2o = torch.cat([R, t], -1)
    world = (cam[:, :, None, :] * c2o[:, None, :3, :3]).sum(-1)
    unit = world / torch.norm(world, dim=-1, keepdim=True)
    origin = c2o[:, None, :3, 3].expand(world.shape)
    dist = c2o[:, :, 3].detach().norm(dim=-1)
    near, far = (dist - half)[:, None], (dist + half)[:, None]
    idx = torch.arange(S, dtype=cam.dtype)[None, :]
    hw = (far - near) / (2 * S)
    start, end = near + hw, far - hw
    step = (end - start) / max(S - 1, 1)
    z = torch.where(idx < S // 2, start + step * idx, end - step * (S - 1 - idx)) + jit * hw
    return c2o, origin.reshape(-1, 3), unit.reshape(-1, 3), z


@pytest.mark.parametrize("opt_cam_pose", [0, 1])
@pytest.mark.parametrize("B,n,S", [(1, 4096, 64), (3, 100, 32), (2, 1, 7)])
def test_pose_rays_forward_backward(amd, dev, B, n, S, opt_cam_pose):
    D, U = amd.driver, amd.utils
    g = torch.Generator().manual_seed(7 + B + n)
    rot = torch.randn(B, 3, generator=g, dtype=torch.float64) * 1.2
    tr = torch.randn(B, 3, generator=g, dtype=torch.float64) * 3 + torch.tensor([0., 1., 15.], dtype=torch.float64)
    cam = torch.cat([torch.randn(B, n, 2, generator=g, dtype=torch.float64) * 0.2, torch.ones(B, n, 1, dtype=torch.float64)], -1)
    half = torch.rand(B, generator=g, dtype=torch.float64) + 2
    jit = torch.rand(B, S, generator=g, dtype=torch.float64)
    w_o, w_d, w_c = [torch.randn(*s, generator=g, dtype=torch.float64) for s in ((B * n, 3), (B * n, 3), (B, 3, 4))]
    rot_r, tr_r = rot.clone().requires_grad_(), tr.clone().requires_grad_()
    ref = torch_pose_rays(D, U, rot_r, tr_r, cam, half, jit, S, opt_cam_pose)         # float64 reference of the driver's formulas
    ((ref[1] * w_o).sum() + (ref[2] * w_d).sum() + (ref[0] * w_c).sum()).backward()
    f = lambda t: t.float().to(dev)
    rot_d, tr_d = f(rot).requires_grad_(), f(tr).requires_grad_()
    out = amd.ops.PoseRays.apply(rot_d, tr_d, f(cam), f(half), f(jit), S, opt_cam_pose)
    for a, b, name, tol in zip(out, ref, ("cam2opt", "rays_o", "viewdir", "z"), (2e-6, 2e-6, 5e-7, 5e-6)):
        assert md(a, b) < tol * max(1.0, float(b.abs().max())), (name, md(a, b))
    assert not out[3].requires_grad
    ((out[1] * f(w_o)).sum() + (out[2] * f(w_d)).sum() + (out[0] * f(w_c)).sum()).backward()
    assert rel(rot_d.grad, rot_r.grad) < 5e-5 and rel(tr_d.grad, tr_r.grad) < 5e-5, (rel(rot_d.grad, rot_r.grad), rel(tr_d.grad, tr_r.grad))


def test_pose_rays_small_angle(amd, dev):
    """Below |v|^2 = 1e-8 the rotation uses its series (driver.axis_angle_to_matrix); value and gradient stay finite and right."""
    D, U = amd.driver, amd.utils
    rot = torch.tensor([[2e-5, -1e-5, 3e-5], [0.0, 0.0, 0.0]], dtype=torch.float64)
    tr = torch.tensor([[0.5, 1.0, 12.0], [1.0, -1.0, 9.0]], dtype=torch.float64)
    g = torch.Generator().manual_seed(1)
    cam = torch.cat([torch.randn(2, 50, 2, generator=g, dtype=torch.float64) * 0.2, torch.ones(2, 50, 1, dtype=torch.float64)], -1)
    half, jit = torch.tensor([2.5, 2.7], dtype=torch.float64), torch.rand(2, 16, generator=g, dtype=torch.float64)
    w_d = torch.randn(100, 3, generator=g, dtype=torch.float64)
    rot_r, tr_r = rot.clone().requires_grad_(), tr.clone().requires_grad_()
    ref = torch_pose_rays(D, U, rot_r, tr_r, cam, half, jit, 16, 0)
    (ref[2] * w_d).sum().backward()
    f = lambda t: t.float().to(dev)
    rot_d, tr_d = f(rot).requires_grad_(), f(tr).requires_grad_()
    out = amd.ops.PoseRays.apply(rot_d, tr_d, f(cam), f(half), f(jit), 16, 0)
    (out[2] * f(w_d)).sum().backward()
    assert md(out[2], ref[2]) < 5e-7 and rel(rot_d.grad, rot_r.grad) < 5e-5 and bool(torch.isfinite(rot_d.grad).all())


# ------------------------------------------------------------------ metric row (src/optimizer_nuscenes.py:739-765, src/utils.py:675-722)
@pytest.mark.parametrize("opt_cam_pose", [0, 1])
def test_metric_row(amd, dev, opt_cam_pose):
    D = amd.driver
    g = torch.Generator().manual_seed(3)
    B, nl = 4, 37
    R = D.axis_angle_to_matrix(torch.randn(B, 3, generator=g)); t = torch.randn(B, 3, 1, generator=g) * 5
    c2o = torch.cat([R, t], -1)
    gtR = D.axis_angle_to_matrix(torch.randn(B, 3, generator=g)); gtT = torch.randn(B, 3, generator=g) * 5
    loss_out = torch.rand(B, 4, generator=g) * 0.2 + 0.01
    d_vec, d0 = torch.rand(B, nl, generator=g) * 20, torch.rand(B, nl, generator=g) * 20
    pred_R = c2o[:, :, :3] if opt_cam_pose else c2o[:, :, :3].transpose(-2, -1)
    pred_t = c2o[:, :, 3:] if opt_cam_pose else -pred_R @ c2o[:, :, 3:]
    want = torch.stack([-10 * torch.log10(loss_out[:, 3]), (d_vec - d0).abs().mean(dim=1), D.rot_dist(pred_R, gtR),
                        (pred_t - gtT[:, :, None]).flatten(1).norm(dim=1)], dim=1)
    f = lambda t_: t_.to(dev).contiguous()
    row, d0_d = torch.zeros(B, 4, device=dev), f(d0)
    amd.ops.metric_row(f(loss_out), f(d_vec), d0_d, False, f(c2o), f(gtR), f(gtT), opt_cam_pose, row)
    assert md(row[:, :2], want[:, :2]) < 2e-5 and md(row[:, 2], want[:, 2]) < 1e-3 and md(row[:, 3], want[:, 3]) < 1e-5, (row.cpu(), want)
    # acos near 0 / pi amplifies the last bits of the trace: compare the cosines there
    assert md(torch.cos(row[:, 2]), torch.cos(want[:, 2])) < 2e-6
    amd.ops.metric_row(f(loss_out), f(d_vec), d0_d, True, f(c2o), f(gtR), f(gtT), opt_cam_pose, row)       # first iteration: depth0 <- depth
    assert float(row[:, 1].abs().max()) == 0.0 and torch.equal(d0_d.cpu(), d_vec)
    # per-object depth-pixel counts (src/optimizer_nuscenes.py:1736-1741: sum |d - gt| / (len + 1e-8) over THAT object's lidar returns)
    cnt = torch.tensor([37, 5, 0, 20], dtype=torch.int32)
    want_d = torch.stack([(d_vec[b, :int(c)] - d0[b, :int(c)]).abs().sum() / (int(c) + 1e-8) for b, c in enumerate(cnt)])
    amd.ops.metric_row(f(loss_out), f(d_vec), f(d0), False, f(c2o), f(gtR), f(gtT), opt_cam_pose, row, lidar_count=cnt.to(dev))
    assert md(row[:, 1], want_d) < 2e-5 and float(row[2, 1]) == 0.0


# ------------------------------------------------------------------ per-object latent layers (src/model_supnerf.py:253,261)
@pytest.mark.parametrize("sb,tb,B", [(3, 1, 1), (3, 1, 5), (2, 1, 64), (1, 0, 3), (0, 2, 2), (5, 5, 2)])
def test_latent_layers_one_launch_matches_torch(amd, dev, sb, tb, B):
    """z_j = ReLU(Lin_j(code)) for every block and the biases they fold into, one launch, against the per-layer nn.Linear form in float64;
    the gradient to both codes against torch's autograd of that form (a code no layer reads gets zeros)."""
    torch.manual_seed(sb * 10 + tb)
    m = amd.CodeNeRF(sb, tb).to(dev)
    g = torch.Generator().manual_seed(B)
    sc0, tc0 = torch.randn(B, 256, generator=g) * 0.3, torch.randn(B, 256, generator=g) * 0.3
    up = torch.randn(B, sb + tb, 256, generator=g)
    sc, tc = sc0.to(dev).requires_grad_(), tc0.to(dev).requires_grad_()
    z = m.latent_terms(sc, tc)
    lb = m.latent_biases(z)
    assert getattr(z, "_snr_latent_bias", None) is lb                     # (the one-launch path ran)
    (z * up.to(dev)).sum().backward()
    m64 = amd.CodeNeRF(sb, tb).double(); m64.load_state_dict({k: v.double().cpu() for k, v in m.state_dict().items()})
    s64, t64 = sc0.double().requires_grad_(), tc0.double().requires_grad_()
    lat = [getattr(m64, f"shape_latent_layer_{j + 1}")(s64) for j in range(sb)] + [getattr(m64, f"texture_latent_layer_{j + 1}")(t64) for j in range(tb)]
    nxt = [getattr(m64, f"shape_layer_{j + 1}")[0] for j in range(sb)] + [getattr(m64, f"texture_layer_{j + 1}")[0] for j in range(tb)]
    z64 = torch.stack(lat, 1)
    lb64 = torch.stack([lin(z64[:, j]) for j, lin in enumerate(nxt)], 1)
    (z64 * up.double()).sum().backward()
    assert md(z, z64) < 2e-6 and md(lb, lb64) < 5e-6, (md(z, z64), md(lb, lb64))
    for got, want in ((sc.grad, s64.grad), (tc.grad, t64.grad)):
        want = torch.zeros(B, 256, dtype=torch.float64) if want is None else want
        assert md(got, want) < 1e-5 * max(1.0, float(want.abs().max())), md(got, want)


# ------------------------------------------------------------------ AdamW (src/optimizer_nuscenes.py:1762-1769: torch.optim.AdamW defaults)
def test_device_adamw_matches_torch(amd, dev):
    g = torch.Generator().manual_seed(11)
    shapes, lrs = [(2, 256), (2, 256), (2, 3), (2, 3)], [0.02, 0.015, 0.01, 0.01]
    p_ref = [torch.randn(*s, generator=g).requires_grad_() for s in shapes]
    p_dev = [p.detach().clone().to(dev).requires_grad_() for p in p_ref]
    ref = torch.optim.AdamW([{"params": p, "lr": lr} for p, lr in zip(p_ref, lrs)], foreach=False)
    mine = amd.ops.DeviceAdamW(list(zip(p_dev, lrs)))
    for step in range(25):
        for a, b in zip(p_ref, p_dev):
            gr = torch.randn(*a.shape, generator=g) * (0.1 + step % 3)
            a.grad, b.grad = gr.clone(), gr.to(dev)
        ref.step(); mine.step()
        if step == 11:      # what re-creating the optimiser with halved rates does
            mine.restart(0.5)
            ref = torch.optim.AdamW([{"params": p, "lr": lr * 0.5} for p, lr in zip(p_ref, lrs)], foreach=False)
    for a, b in zip(p_ref, p_dev):
        assert md(b, a) < 2e-6, md(b, a)


def test_table_adamw_matches_torch_and_bumps_versions(amd, dev):
    """The training step's optimiser (src/trainer_unified_nuscenes.py:414-422): ONE launch for any number of tensors in up to four groups,
    gradients read from the fixed buffers the parameters held when it was built (bucket views at odd offsets of one flat buffer), the
    parameters' version counters bumped so that caches keyed on them notice.  Against torch.optim.AdamW on the CPU, 25 steps."""
    g = torch.Generator().manual_seed(12)
    shapes = [(256, 63), (256,), (256, 256), (1, 256), (1,), (256, 283), (3, 128), (3,), (6, 256), (6, 256), (70001,)]
    group = [0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 0]
    lrs = [1e-2, 2e-2, 5e-3]
    p_ref = [torch.randn(*s, generator=g).requires_grad_() for s in shapes]
    p_dev = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in p_ref]
    flat = torch.zeros(sum(p.numel() for p in p_dev), device=dev)
    off = 0
    for p in p_dev:                                   # gradient buffers = views into one flat buffer (GradBucket's layout)
        p.grad = flat[off:off + p.numel()].view_as(p); off += p.numel()
    ref = torch.optim.AdamW([{"params": [p for p, k in zip(p_ref, group) if k == gi], "lr": lr} for gi, lr in enumerate(lrs)], foreach=False)
    mine = amd.ops.TableAdamW([([p for p, k in zip(p_dev, group) if k == gi], lr) for gi, lr in enumerate(lrs)])
    v0 = [p._version for p in p_dev]
    for step in range(25):
        for a, b in zip(p_ref, p_dev):
            gr = torch.randn(*a.shape, generator=g) * (0.1 + step % 3)
            a.grad = gr.clone(); b.grad.copy_(gr.to(dev))
        ref.step(); mine.step()
    for a, b in zip(p_ref, p_dev):
        assert md(b, a) < 2e-6, (tuple(a.shape), md(b, a))
    assert all(p._version > v for p, v in zip(p_dev, v0))
    with pytest.raises(amd.SnrError):
        mine.zero_grad(set_to_none=True)
    mine.zero_grad()
    assert float(flat.abs().max()) == 0.0


# ------------------------------------------------------------------ weight gradients (src/trainer_unified_nuscenes.py:334: the dW half of backward)
@pytest.mark.parametrize("P,n_out,ldg,n_in,ldx", [(4096, 256, 256, 256, 256), (70001, 128, 256, 256, 256), (1537, 256, 256, 64, 64), (999, 256, 256, 28, 28),
                                                  (20000, 3, 3, 128, 256), (20000, 1, 1, 256, 256), (2, 256, 256, 256, 256), (524288, 256, 256, 256, 256)])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_weight_grad_matches_matmul(amd, dev, P, n_out, ldg, n_in, ldx, precision):
    """dW = G^T X and db = column sums of G on the split-K fp32-MFMA kernel against a float64 matmul; ragged point counts (slices that end
    mid k-step), narrow heads, operands that are column blocks of wider buffers, writing into a column block of a wider dW."""
    g = torch.Generator().manual_seed(P % 1000 + n_out)
    G = torch.randn(P, ldg, generator=g) * 0.1
    X = torch.randn(P, ldx, generator=g)
    want_w = (G[:, :n_out].double().t() @ X[:, :n_in].double())
    want_b = G[:, :n_out].double().sum(0)
    Gd, Xd = G.to(dev), X.to(dev)
    dW, db = amd.ops.weight_grad(Gd[:, :n_out] if ldg > n_out else Gd, n_out, Xd, n_in, precision=precision)
    scale = float(want_w.abs().max())
    # fp32: accumulation round-off only; bf16x3: + the dropped lo*lo terms, 2^-17 per product, random signs over P points
    tol = (3e-6 if precision == "fp32" or n_out < 32 else 2e-5) * scale * max(1.0, (P / 4096) ** 0.5)
    assert dW.shape == (n_out, n_in) and md(dW, want_w) < tol, (md(dW, want_w), scale)
    assert md(db, want_b) < 3e-6 * float(want_b.abs().max() + 1) * max(1.0, (P / 4096) ** 0.5)
    wide = torch.full((n_out, n_in + 12), 7.0, device=dev)          # a column block of a wider matrix; the rest stays untouched
    amd.ops.weight_grad(Gd, n_out, Xd, n_in, out=(wide[:, 4:4 + n_in], None), precision=precision)
    assert torch.equal(wide[:, 4:4 + n_in], dW) and float(wide[:, :4].min()) == 7.0 and float(wide[:, 4 + n_in:].max()) == 7.0
    dW2, _ = amd.ops.weight_grad(Gd, n_out, Xd, n_in, precision=precision)                # deterministic: the same bits every time
    assert torch.equal(dW2, dW)


def test_weight_grad_rejects_misuse(amd, dev):
    with pytest.raises(amd.SnrError):
        amd.ops.weight_grad(torch.rand(8, 256), 256, torch.rand(8, 256), 256)                                        # CPU tensors
    with pytest.raises(amd.SnrError):
        amd.ops.weight_grad(torch.rand(8, 256, device=dev), 256, torch.rand(9, 256, device=dev), 256)                # different point counts
    with pytest.raises(amd.SnrError):
        amd.ops.weight_grad(torch.rand(8, 63, device=dev), 63, torch.rand(8, 256, device=dev), 256)                  # 63 rows: not a multiple of 4
    with pytest.raises(amd.SnrError):
        amd.ops.weight_grad(torch.rand(8, 16, device=dev), 16, torch.rand(8, 256, device=dev), 256)                  # 5..31 rows: no such layer


# ------------------------------------------------------------------ the fused loop == the loop on the public API
@pytest.mark.parametrize("precision", ["fp32", "auto"])
def test_fused_loop_equals_api_loop(amd, dev, oracle_params, precision):
    """optimize_object (about thirty launches per iteration) against optimize_object_api (the reference's call sequence on the public
    functions): same jitter, same start -> same traces up to fp32 round-off through Adam."""
    D = amd.driver
    model = amd.CodeNeRF(3, 1); model.load_state_dict(oracle_params); model.precision = precision; model = model.to(dev)
    hp = D.load_hpams(); hp["render_im_sz"] = 16; hp["optimize"]["num_opts"] = 10; hp["optimize"]["lr_half_interval"] = 6
    obj = D.make_objects([21], 16)[0]
    g = torch.Generator().manual_seed(5)
    sc0, tc0 = torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3
    jit = torch.rand(10, 2, 64, generator=g)
    a = D.optimize_object_api(model, dev, obj, hp, sc0, tc0, reg_iters=1, seed=9, jitter=jit)
    b = D.optimize_object(model, dev, obj, hp, sc0, tc0, reg_iters=1, seed=9, jitter=jit)
    assert all(p.requires_grad for p in model.parameters())                       # the loop thaws what it froze
    d = (a[0] - b[0]).abs()
    print(f"[fused vs api loop, {precision}] max metric differences {d.max(dim=0)[0].tolist()}")
    assert float(d[:3].max()) < 2e-4, d[:3]                                       # before the first optimiser step: the same numbers
    # afterwards fp32 round-off through Adam's normalisation (and its restart at the lr halving); the depth column is the rendered depth of
    # near-empty space at random-init density, the most sensitive quantity of the four
    assert float(d[:, 0].max()) < 0.05 and float(d[:, 1].max()) < 5e-2 and float(d[:, 2].max()) < 2e-3 and float(d[:, 3].max()) < 5e-2, d
    assert md(a[3], b[3]) < 5e-2 and md(a[1], b[1]) < 5e-2
    # one optimiser step, compared sharply: both loops stopped right after their first AdamW update
    hp1 = D.load_hpams(); hp1["render_im_sz"] = 16; hp1["optimize"]["num_opts"] = 3
    a1 = D.optimize_object_api(model, dev, obj, hp1, sc0, tc0, reg_iters=1, seed=9, jitter=jit[:3])
    b1 = D.optimize_object(model, dev, obj, hp1, sc0, tc0, reg_iters=1, seed=9, jitter=jit[:3])
    # (Adam's first step is lr * g / (|g| + eps): where |g| ~ eps the two loops' round-off decides the step, hence the 1e-3 of lr = 0.02)
    assert md(a1[1], b1[1]) < 2e-3 and md(a1[2], b1[2]) < 2e-3 and float((a1[0] - b1[0]).abs().max()) < 2e-4
    # the global-RNG jitter stream: two torch.rand(S) per iteration, in order, like the reference's loop
    torch.manual_seed(77); a2 = D.optimize_object_api(model, dev, obj, hp, sc0, tc0, reg_iters=1, seed=9)
    torch.manual_seed(77); b2 = D.optimize_object(model, dev, obj, hp, sc0, tc0, reg_iters=1, seed=9)
    assert float((a2[0][:3] - b2[0][:3]).abs().max()) < 2e-4


# ------------------------------------------------------------------ public get_rays / depths as one launch (src/utils.py:107-135,159-164,468-469)
@pytest.mark.parametrize("im_sz,subset", [(64, None), (24, None), (33, 700), (5, None)])
def test_public_rays_and_depths_one_launch(amd, dev, im_sz, subset):
    """``utils.get_rays`` and the rays + depth vector of ``render_rays_v2`` with the pose on the GPU (``snr_cam_rays_fwd/bwd``) against the
    torch formulation the same functions run for a CPU pose (the reference's own lines), values and the gradient wrt the pose; ray
    counts that leave partial 1024-ray chunks and a ray subset included."""
    U = amd.utils
    ob = O.synthetic_object(3)
    S = 64
    g = torch.Generator().manual_seed(im_sz)
    jit = torch.rand(S, generator=g)
    ids = None if subset is None else np.random.RandomState(0).permutation(im_sz * im_sz)[:subset]
    n = im_sz * im_sz if ids is None else subset
    w_o, w_d = torch.rand(n, 3, generator=g, dtype=torch.float64), torch.rand(n, 3, generator=g, dtype=torch.float64)

    pose64 = ob["cam_pose"].double().requires_grad_()
    ro64, vd64 = O.pixel_rays(ob["K"].double(), pose64, ob["roi"], uv_steps=[im_sz, im_sz])
    if ids is not None:
        ro64, vd64 = ro64[ids], vd64[ids]
    ((ro64 * w_o).sum() + (vd64 * w_d).sum()).backward()
    near, far = O.sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z_ref = O.shared_depth_samples(near, far, S, jit)

    pose_d = ob["cam_pose"].to(dev).requires_grad_()
    U.JITTER_OVERRIDE = jit
    try:
        ro, vd, z = U._rays_and_depths(ob["K"], pose_d, ob["roi"], [im_sz, im_sz], ob["obj_diag"], S, ids=ids)
    finally:
        U.JITTER_OVERRIDE = None
    assert ro.is_contiguous() and ro.shape == (n, 3) and z.shape == (S,)
    assert md(ro, ro64) < 1e-6 and md(vd, vd64) < 2e-7 and md(z, z_ref) < 4e-6
    ro32, vd32 = O.pixel_rays(ob["K"], ob["cam_pose"], ob["roi"], uv_steps=[im_sz, im_sz])       # the reference's fp32 rays, bit for bit
    if ids is not None:
        ro32, vd32 = ro32[ids], vd32[ids]
    assert md(ro, ro32) == 0.0 and md(vd, vd32) == 0.0
    ((ro * w_o.float().to(dev)).sum() + (vd * w_d.float().to(dev)).sum()).backward()
    assert rel(pose_d.grad, pose64.grad) < 2e-5
    # the bare public function takes the same path
    if ids is None:
        ro2, vd2 = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[im_sz, im_sz])
        assert torch.equal(ro2, ro.detach()) and torch.equal(vd2, vd.detach())
