"""Parity of the HIP kernels (through the C ABI) against the committed golden vectors and the CPU oracle.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O

pytestmark = pytest.mark.gpu

# fp32 tolerances.  BASELINE.json north_star: PSNR delta <= 0.01 dB, depth L1 <= 1e-4 (mean abs, metres).
# The kernels compute in fp32 (fp32 MFMA == fmaf chain), so they are held to a much tighter bar.
TOL_RGB = 2e-5
TOL_DEPTH_MEAN = 1e-5      # metres, mean abs  (north_star bound: 1e-4)
TOL_DEPTH_MAX = 1e-4
TOL_ACC = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def amd():
    import supnerf_amd
    return supnerf_amd


@pytest.fixture(scope="module")
def packed(amd, dev, oracle_params):
    p = {k: v.to(dev) for k, v in oracle_params.items()}
    return amd.ops.pack_weights(p, 3, 1), p


def maxdiff(a, b):
    return float((a.detach().double().cpu() - torch.as_tensor(b).double().cpu()).abs().max())


# ------------------------------------------------------------------ composite
def test_composite_variants(amd, dev, golden):
    g = golden("composite")
    ops = amd.ops
    sig, rgbs = g["sigmas"].to(dev), g["rgbs"].to(dev)
    out = ops.composite_fwd(sig.squeeze(-1), rgbs, g["z_shared"].to(dev), ops.Z_SHARED, False)
    for a, k in zip(out, ("vr2_rgb", "vr2_depth", "vr2_acc")):
        assert maxdiff(a, g[k]) < 2e-5, k
    out = ops.composite_fwd(sig.squeeze(-1), rgbs, g["z_ray"].to(dev), ops.Z_PER_RAY, True)
    for a, k in zip(out, ("white_rgb", "white_depth", "white_acc")):
        assert maxdiff(a, g[k]) < 2e-5, k
    out = ops.composite_fwd(sig.squeeze(-1), rgbs, g["z_ray"].to(dev), ops.Z_PER_RAY, False)
    for a, k in zip(out, ("vr3_rgb", "vr3_depth", "vr3_acc")):
        assert maxdiff(a, g[k]) < 2e-5, k
    B, S = g["z_obj"].shape
    out = ops.composite_fwd(sig.squeeze(-1), rgbs, g["z_obj"].to(dev), ops.Z_PER_OBJECT, False, rays_per_obj=sig.shape[0] // B)
    for a, k in zip(out, ("batch_rgb", "batch_depth", "batch_acc")):
        assert maxdiff(a.view(g[k].shape), g[k]) < 2e-5, k


def test_composite_backward(amd, dev, golden):
    g = golden("composite_grad")
    ops = amd.ops
    sig = g["sigmas"].squeeze(-1).to(dev).requires_grad_()
    rgbs = g["rgbs"].to(dev).requires_grad_()
    z = g["z"].to(dev).requires_grad_()
    r = ops.Composite.apply(sig, rgbs, z, ops.Z_PER_RAY, True, 0)
    assert maxdiff(r[0], g["rgb"]) < 2e-5 and maxdiff(r[1], g["depth"]) < 2e-5 and maxdiff(r[2], g["acc"]) < 2e-5
    ((r[0] * g["w_rgb"].to(dev)).sum() + (r[1] * g["w_depth"].to(dev)).sum() + (r[2] * g["w_acc"].to(dev)).sum()).backward()
    assert maxdiff(sig.grad, g["d_sigmas"].squeeze(-1)) < 1e-4
    assert maxdiff(rgbs.grad, g["d_rgbs"]) < 2e-5
    assert maxdiff(z.grad, g["d_z"]) < 2e-4


@pytest.mark.parametrize("S", [2, 7, 64, 65, 130, 256])   # S=1 is invalid in the reference too (empty delta)
def test_composite_ragged_sample_counts(amd, dev, S):
    ops = amd.ops
    gen = torch.Generator().manual_seed(S)
    N = 9
    sig = (torch.rand(N, S, generator=gen) * 2).requires_grad_()
    rgbs = torch.randn(N, S, 3, generator=gen).requires_grad_()
    z = torch.sort(torch.rand(N, S, generator=gen) * 3 + 4, dim=-1)[0].requires_grad_()
    w = [torch.randn(N, 3, generator=gen), torch.randn(N, generator=gen), torch.randn(N, generator=gen)]
    ref = O.composite(sig, rgbs, z, white_bkgd=True)
    sum(((a * b).sum() for a, b in zip(ref, w))).backward()
    sg, rg, zg = [t.detach().to(dev).requires_grad_() for t in (sig, rgbs, z)]
    out = ops.Composite.apply(sg, rg, zg, ops.Z_PER_RAY, True, 0)
    sum(((a * b.to(dev)).sum() for a, b in zip(out, w))).backward()
    for a, b in zip(out, ref):
        assert maxdiff(a, b) < 3e-5
    assert maxdiff(sg.grad, sig.grad) < 2e-4
    assert maxdiff(rg.grad, rgbs.grad) < 3e-5
    assert maxdiff(zg.grad, z.grad) < 5e-4


def test_composite_empty(amd, dev):
    ops = amd.ops
    out = ops.composite_fwd(torch.empty(0, 64, device=dev), torch.empty(0, 64, 3, device=dev), torch.rand(64, device=dev), ops.Z_SHARED, False)
    assert out[0].shape == (0, 3) and out[1].shape == (0,)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("P", [32, 64, 96, 160, 224])
def test_partial_workgroups_stay_inside_their_buffers(amd, dev, packed, oracle_params, P, precision):
    """A workgroup owns four 32-point wave tiles; when the launch ends inside a workgroup the tiles past the end must neither
    store ReLU bits nor latent-gradient partials (buffers are sized for ceil(P/32) tiles) -- checked with canaries behind the
    two buffers, through the C ABI with caller-owned memory -- and the results must still match the oracle."""
    import ctypes as C
    lib = amd._lib.lib()
    pk, p_dev = packed
    gen = torch.Generator().manual_seed(P)
    xyz = (torch.rand(P, 1, 3, generator=gen) - 0.5)
    vd = torch.nn.functional.normalize(torch.randn(P, 1, 3, generator=gen), dim=-1)
    sc, tc = torch.randn(1, 256, generator=gen) * 0.3, torch.randn(1, 256, generator=gen) * 0.3
    lat = O.latent_terms(oracle_params, sc, tc).clone().requires_grad_()
    # oracle: decoder on explicit latent terms via the codes (latent layers are linear+relu of the codes; compare d latent through them)
    sc_r, tc_r = sc.clone().requires_grad_(), tc.clone().requires_grad_()
    sig_o, rgb_o = O.decoder_forward(oracle_params, xyz, vd, sc_r, tc_r)
    w_s, w_c = torch.randn(P, generator=gen), torch.randn(P, 3, generator=gen)
    ((sig_o.view(P) * w_s).sum() + (rgb_o.view(P, 3) * w_c).sum()).backward()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    x_d, v_d, lat_d = xyz.view(P, 3).to(dev).contiguous(), vd.view(P, 3).to(dev).contiguous(), lat.detach().to(dev).contiguous()
    prec = 0 if precision == "fp32" else 1
    mbytes = lib.snr_mask_bytes(P, 3, 1)
    masks = torch.full((mbytes + 65536,), 0xAB, dtype=torch.uint8, device=dev)
    sig, rgb = torch.empty(P, device=dev), torch.empty(P, 3, device=dev)
    assert lib.snr_decoder_fwd(ptr(x_d), ptr(v_d), ptr(lat_d), ptr(pk), P, P, 3, 1, ptr(sig), ptr(rgb), ptr(masks), None, prec, st) == 0
    torch.cuda.synchronize()
    assert bool((masks[mbytes:] == 0xAB).all()), "ReLU bits were stored past the end of the mask buffer"
    assert maxdiff(sig, sig_o.view(P)) < 2e-5 and maxdiff(rgb, rgb_o.view(P, 3)) < 2e-5
    wsb = lib.snr_decoder_bwd_ws_bytes(P, P, 3, 1)
    ws = torch.full((wsb + 65536,), 0xCD, dtype=torch.uint8, device=dev)
    d_lat, d_x, d_v = torch.empty_like(lat_d), torch.empty(P, 3, device=dev), torch.empty(P, 3, device=dev)
    assert lib.snr_decoder_bwd(ptr(x_d), ptr(v_d), ptr(lat_d), ptr(pk), ptr(masks), ptr(sig), ptr(w_s.to(dev)), ptr(w_c.to(dev).contiguous()), P, P, 3, 1,
                               ptr(d_lat), ptr(d_x), ptr(d_v), None, ptr(ws), wsb, prec, st) == 0
    torch.cuda.synchronize()
    assert bool((ws[wsb:] == 0xCD).all()), "latent-gradient partials were stored past the end of the workspace"
    # d latent -> d codes through the (tiny, stock) latent layers, to compare with the oracle's code gradients
    sc_g, tc_g = sc.clone().requires_grad_(), tc.clone().requires_grad_()
    O.latent_terms(oracle_params, sc_g, tc_g).backward(d_lat.cpu())
    # mask-matched (tests/relu_bits.py): the oracle differentiates with the ReLU bits this launch saved, so a hidden unit within rounding of
    # its kink cannot move a tile's gradient by percents any more, and the bound is 1e-4 for both arithmetics (rounds 1-2: 5e-2 for split-bf16)
    from relu_bits import decode_relu_bits
    sc_m, tc_m = sc.clone().requires_grad_(), tc.clone().requires_grad_()
    sig_m, rgb_m = O.decoder_forward(oracle_params, xyz, vd, sc_m, tc_m, relu_masks=decode_relu_bits(masks[:mbytes], P, 3, 1))
    ((sig_m.view(P) * w_s).sum() + (rgb_m.view(P, 3) * w_c).sum()).backward()
    assert maxdiff(sc_g.grad, sc_m.grad) <= 1e-4 * float(sc_m.grad.abs().max()) + 1e-7
    assert maxdiff(tc_g.grad, tc_m.grad) <= 1e-4 * float(tc_m.grad.abs().max()) + 1e-7


def test_scene_composite_golden(amd, dev, golden):
    """Per-pixel depth merge + white-background composite (scripts/demo.py:555-565) on the reference's own batch."""
    g = golden("scene")
    n_obj = int(g["obj_poses"].shape[0])
    S = g["b0_z"].shape[1] // n_obj
    for run in (0, S):          # rank sort, and the merge of the per-object sorted lists (what scene.py asks for)
        rgb, depth, acc = amd.ops.scene_composite(g["b0_sigmas"].to(dev), g["b0_rgbs"].to(dev), g["b0_z"].to(dev), run_length=run)
        assert maxdiff(rgb, g["b0_rgb"]) < TOL_RGB and maxdiff(acc, g["b0_acc"]) < TOL_ACC
        assert float((depth.cpu() - g["b0_depth"]).abs().mean()) < TOL_DEPTH_MEAN and maxdiff(depth, g["b0_depth"]) < TOL_DEPTH_MAX


@pytest.mark.parametrize("Nb,S,P", [(1, 64, 37), (3, 64, 501), (4, 64, 257), (2, 128, 40), (4, 32, 77), (8, 32, 33), (5, 64, 130), (8, 64, 64), (7, 33, 50),
                                     (16, 64, 9), (2, 5, 1000)])
def test_scene_composite_shapes(amd, dev, Nb, S, P):
    """Other object counts incl. samples per pixel that are no multiple of 64, empty objects (depth -1) and exact ties
    between objects; checked against the oracle's restatement of the reference (sort + searchsorted scatter).  Up to 256 samples per pixel in
    lists of 32 / 64 / 128 take the fast kernel + the general kernel for the pixels it marks (ties), everything else the general kernel alone."""
    gen = torch.Generator().manual_seed(Nb * 1000 + S)
    near = torch.rand(P, Nb, 1, generator=gen) * 20 + 2
    z = near + torch.sort(torch.rand(P, Nb, S, generator=gen), dim=-1)[0] * 4
    sig = torch.rand(P, Nb, S, generator=gen) * 2 - 0.3                       # some negative densities (relu in the composite)
    rgb = torch.rand(P, Nb, S, 3, generator=gen)
    empty = torch.rand(P, Nb, generator=gen) < 0.3                            # objects that miss the pixel
    z[empty] = -1; sig[empty] = 0; rgb[empty] = 1
    z, sig, rgb = z.view(P, Nb * S), sig.view(P, Nb * S), rgb.view(P, Nb * S, 3)
    want = O.scene_composite(sig, rgb, z)
    generic = amd.ops.scene_composite(sig.to(dev), rgb.to(dev), z.to(dev))
    for run in (0, S):
        got = amd.ops.scene_composite(sig.to(dev), rgb.to(dev), z.to(dev), run_length=run)
        assert maxdiff(got[0], want[0]) < TOL_RGB and maxdiff(got[2], want[2]) < TOL_ACC
        assert maxdiff(got[1], want[1]) < TOL_DEPTH_MAX
        assert all(torch.equal(a, b) for a, b in zip(got, generic))             # the merge path places every sample where the rank sort does
    if Nb > 1:      # two objects sharing every depth exactly: the reference keeps the later sample of each pair and drops the other
        z2 = z.clone().view(P, Nb, S); z2[:, 1] = z2[:, 0]; z2 = z2.view(P, Nb * S)
        want2 = O.scene_composite(sig, rgb, z2)
        for run in (0, S):
            got2 = amd.ops.scene_composite(sig.to(dev), rgb.to(dev), z2.to(dev), run_length=run)
            assert maxdiff(got2[0], want2[0]) < TOL_RGB and maxdiff(got2[1], want2[1]) < TOL_DEPTH_MAX and maxdiff(got2[2], want2[2]) < TOL_ACC
    # repeated depths INSIDE a list (ties within one object and across objects), and a wrong hint: lists that are not ascending must give the
    # rank sort's answer (the kernel checks the order per pixel)
    z3 = (z.view(P, Nb, S) * 4).round() / 4
    z3 = z3.view(P, Nb * S)
    want3 = O.scene_composite(sig, rgb, z3)
    got3 = amd.ops.scene_composite(sig.to(dev), rgb.to(dev), z3.to(dev), run_length=S)
    assert maxdiff(got3[0], want3[0]) < TOL_RGB and maxdiff(got3[1], want3[1]) < TOL_DEPTH_MAX and maxdiff(got3[2], want3[2]) < TOL_ACC
    zr = z.view(P, Nb, S).flip(-1).reshape(P, Nb * S).contiguous()
    want4 = O.scene_composite(sig, rgb, zr)
    got4 = amd.ops.scene_composite(sig.to(dev), rgb.to(dev), zr.to(dev), run_length=S)
    assert maxdiff(got4[0], want4[0]) < TOL_RGB and maxdiff(got4[1], want4[1]) < TOL_DEPTH_MAX and maxdiff(got4[2], want4[2]) < TOL_ACC


def test_scene_composite_limits(amd, dev):
    e = amd.ops.scene_composite(torch.zeros(0, 128, device=dev), torch.zeros(0, 128, 3, device=dev), torch.zeros(0, 128, device=dev))
    assert e[0].shape == (0, 3)
    with pytest.raises(amd.SnrError):
        amd.ops.scene_composite(torch.zeros(2, 2000, device=dev), torch.zeros(2, 2000, 3, device=dev), torch.zeros(2, 2000, device=dev))
    with pytest.raises(amd.SnrError):
        amd.ops.scene_composite(torch.zeros(2, 8, device=dev), torch.zeros(2, 8, 3, device=dev), torch.zeros(2, 9, device=dev))
    with pytest.raises(amd.SnrError):          # a list length that does not divide the samples per pixel
        amd.ops.scene_composite(torch.zeros(2, 8, device=dev), torch.zeros(2, 8, 3, device=dev), torch.zeros(2, 8, device=dev), run_length=3)


# ------------------------------------------------------------------ encode
def _geom_family_a(g, dev, S, shapenet, kitti=False, flip=False):
    ro, vd = O.pixel_rays(g["K"], g["cam_pose"], g["roi"], uv_steps=[int(g["im_sz"])] * 2)
    near, far = O.sphere_bounds(g["cam_pose"], np.float32(g["obj_diag"]))
    z = O.shared_depth_samples(near, far, S, g["jitter"])
    return ro.to(dev), vd.to(dev), z.to(dev)


def frame_matrix(sym_flip=False, kitti2nusc=False, shapenet=False):
    m = np.eye(3, dtype=np.float32)
    if sym_flip:
        m = np.diag([1, -1, 1]).astype(np.float32) @ m
    if kitti2nusc:
        m = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float32) @ m
    if shapenet:
        m = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=np.float32) @ m
    return m.reshape(-1).tolist()


@pytest.mark.parametrize("tag", ["a_nusc", "a_kitti", "a_demo"])
def test_encode_family_a(amd, dev, golden, tag):
    g = golden("render_" + tag)
    ops = amd.ops
    S = int(g["n_samples"])
    ro, vd, z = _geom_family_a(g, dev, S, bool(g["shapenet_obj_cood"]))
    cfg = ops.RenderCfg(S, ops.Z_SHARED, ro.shape[0], 3, 1, frame=frame_matrix(False, bool(g["kitti2nusc"]), bool(g["shapenet_obj_cood"])))
    div = torch.tensor([float(g["obj_diag"])], device=dev)
    xyz, vdir, zz, pe, ped = ops.encode(ro, vd, z, div, None, cfg, want_pe=True)
    xo, vo = O.points_on_rays(ro.cpu(), vd.cpu(), z.cpu())
    xo = xo / np.float32(g["obj_diag"])
    xo, vo = O.object_frame_transforms(xo, vo, False, bool(g["kitti2nusc"]), bool(g["shapenet_obj_cood"]))
    assert maxdiff(xyz, xo) == 0.0
    assert maxdiff(vdir, vo) == 0.0
    assert maxdiff(zz, z.cpu()[None].expand(ro.shape[0], S)) == 0.0
    assert maxdiff(pe, O.positional_encoding(xo, 10)) < 2e-6
    assert maxdiff(ped, O.positional_encoding(vo[:, 0], 4)) < 2e-6


def test_encode_family_b_metric_depth(amd, dev, golden):
    g = golden("render_b_hit")
    ops = amd.ops
    S = int(g["n_samples"])
    wlh = g["wlh"].numpy()
    diag = np.linalg.norm(wlh).astype(np.float32)
    ro, vd = O.pixel_rays(g["K"], g["cam_pose"], g["roi"], uv_steps=[int(g["im_sz"])] * 2)
    xyz_o, vd_o, zv_o, hit = O.aabb_sampled_rays(ro, vd, wlh, S, g["jitter"])
    # per-ray unit depths as the host would compute them
    o_n = ro / (diag / 2)
    tn, tf, h = O.slab_intersect(o_n, vd, -torch.tensor([wlh[1], wlh[0], wlh[2]]) / diag, torch.tensor([wlh[1], wlh[0], wlh[2]]) / diag)
    near = torch.where(h, tn, torch.full_like(tn, -1.0))[:, None]
    far = torch.where(h, tf, torch.full_like(tf, -1.0))[:, None]
    t = O.unit_interval_samples(near, far, S, g["jitter"])
    cfg = ops.RenderCfg(S, ops.Z_PER_RAY, ro.shape[0], 3, 1, metric_z=True)
    one = torch.ones(1, device=dev)
    xyz, vdir, zz = ops.encode(o_n.to(dev), vd.to(dev), t.to(dev), one, torch.tensor([float(diag / 2)], device=dev), cfg)
    assert maxdiff(xyz, xyz_o) < 1e-6
    assert maxdiff(zz, zv_o) < 2e-6
    assert maxdiff(zz, g["z_vals"]) < 2e-6


# ------------------------------------------------------------------ decoder
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("tag", ["b1_s32", "b3_s64", "b2_s7"])
def test_decoder_forward_golden(amd, dev, golden, packed, tag, precision):
    g = golden("decoder_" + tag)
    pk, p = packed
    N, S = g["xyz"].shape[:2]
    lat = g["latent_terms"].to(dev)
    if tag == "b2_s7" and precision == "bf16x3":      # 35 points per object: not whole 32-point tiles -> must refuse loudly
        with pytest.raises(amd.SnrError):
            amd.ops.decoder_fwd(g["xyz"].reshape(-1, 3).to(dev), g["viewdir"].reshape(-1, 3).to(dev), lat, pk, 3, 1, precision=precision)
        return
    sig, rgb, _ = amd.ops.decoder_fwd(g["xyz"].reshape(-1, 3).to(dev), g["viewdir"].reshape(-1, 3).to(dev), lat, pk, 3, 1,
                                      precision=precision)
    assert maxdiff(sig.view(N, S, 1), g["sigmas"]) < 2e-5
    assert maxdiff(rgb.view(N, S, 3), g["rgbs"]) < 2e-5


def test_decoder_forward_other_block_counts(amd, dev):
    """shape_blocks / texture_blocks are run-time parameters (CodeNeRF default 2/1, SUPNeRF default 5/5)."""
    for sb, tb, prec in [(2, 1, "fp32"), (2, 1, "bf16x3"), (5, 5, "fp32"), (0, 0, "fp32"), (0, 0, "bf16x3"), (1, 2, "bf16x3"), (2, 2, "auto"),
                         (5, 5, "auto")]:
        params = O.init_decoder_params(shape_blocks=sb, texture_blocks=tb, seed=3 + sb)
        gen = torch.Generator().manual_seed(sb * 10 + tb)
        N, S, B = 8, 16, 2          # 64 points per object: whole 32-point tiles (needed by bf16x3)
        xyz = torch.rand(N, S, 3, generator=gen) - 0.5
        vd = torch.randn(N, S, 3, generator=gen); vd = vd / vd.norm(dim=-1, keepdim=True)
        sc, tc = torch.randn(B, 256, generator=gen) * 0.3, torch.randn(B, 256, generator=gen) * 0.3
        with torch.no_grad():
            sig_o, rgb_o = O.decoder_forward(params, xyz, vd, sc, tc)
            lat = O.latent_terms(params, sc, tc) if sb + tb else torch.zeros(B, 0, 256)
        pk = amd.ops.pack_weights({k: v.to(dev) for k, v in params.items()}, sb, tb)
        lat_d = lat.to(dev) if sb + tb else torch.zeros(B, 1, 256, device=dev)
        sig, rgb, _ = amd.ops.decoder_fwd(xyz.reshape(-1, 3).to(dev), vd.reshape(-1, 3).to(dev), lat_d, pk, sb, tb, precision=prec)
        assert maxdiff(sig.view(N, S, 1), sig_o) < 2e-5, (sb, tb, prec)
        assert maxdiff(rgb.view(N, S, 3), rgb_o) < 2e-5, (sb, tb, prec)


def test_latent_terms_folded_into_biases(amd, dev, golden, packed, oracle_params):
    """snr_render_args::latent_bias (the latent terms folded into the next layers' biases, b + W z): the split-bf16 forward with it must
    render what it renders with the plain latent terms, and save the same ReLU bits for the backward."""
    g = golden("render_a_nusc")
    ops = amd.ops
    pk, p = packed
    S = int(g["n_samples"])
    ro, vd, z = _geom_family_a(g, dev, S, bool(g["shapenet_obj_cood"]))
    lat = O.latent_terms(oracle_params, g["shapecode"], g["texturecode"]).to(dev)
    div = torch.tensor([float(g["obj_diag"])], device=dev)
    frame = frame_matrix(False, bool(g["kitti2nusc"]), bool(g["shapenet_obj_cood"]))
    names = [f"shape_layer_{j + 1}.0" for j in range(3)] + ["texture_layer_1.0"]
    lb = torch.stack([torch.nn.functional.linear(lat[:, j], oracle_params[n + ".weight"].to(dev), oracle_params[n + ".bias"].to(dev))
                      for j, n in enumerate(names)], dim=1).contiguous()
    outs = []
    for bias in (None, lb):
        cfg = ops.RenderCfg(S, ops.Z_SHARED, ro.shape[0], 3, 1, frame=frame, precision="bf16x3")
        cfg.latent_bias = bias
        outs.append(ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg, save_for_bwd=True))
    for a, b, tol in zip(outs[0][:5], outs[1][:5], (2e-6, 2e-5, 2e-6, 2e-5, 2e-5)):
        assert maxdiff(a, b) < tol
    # ReLU bits: identical up to pre-activations within rounding of zero
    diff_bits = int((outs[0][5] != outs[1][5]).sum())
    assert diff_bits <= outs[0][5].numel() // 100000 + 8, diff_bits
    assert maxdiff(outs[1][0], g["rgb"]) < 2e-5
    with pytest.raises(amd.SnrError):
        cfg.latent_bias = lb[:, :2]
        ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)


# ------------------------------------------------------------------ fused render
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("tag", ["a_nusc", "a_kitti", "a_demo"])
def test_render_family_a_golden(amd, dev, golden, packed, oracle_params, tag, precision):
    g = golden("render_" + tag)
    ops = amd.ops
    pk, p = packed
    S = int(g["n_samples"])
    ro, vd, z = _geom_family_a(g, dev, S, bool(g["shapenet_obj_cood"]))
    lat = O.latent_terms(oracle_params, g["shapecode"], g["texturecode"]).to(dev)
    cfg = ops.RenderCfg(S, ops.Z_SHARED, ro.shape[0], 3, 1, frame=frame_matrix(False, bool(g["kitti2nusc"]), bool(g["shapenet_obj_cood"])),
                        precision=precision)
    div = torch.tensor([float(g["obj_diag"])], device=dev)
    rgb, depth, acc, *_ = ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)
    assert maxdiff(rgb, g["rgb"]) < TOL_RGB
    assert float((depth.cpu() - g["depth"]).abs().mean()) < TOL_DEPTH_MEAN and maxdiff(depth, g["depth"]) < TOL_DEPTH_MAX
    assert maxdiff(acc, g["acc"]) < TOL_ACC


@pytest.mark.parametrize("P", [1, 127, 128, 1000, 70000])
def test_pe_points_layout(amd, dev, P):
    """snr_pe_points: (P,96) = PE(xyz) | 0 | PE(viewdir) | 0 (src/model_supnerf.py:155-161), the X operand of encoding_xyz's and
    encoding_viewdir's weight gradients in the training step; partial blocks included."""
    g = torch.Generator().manual_seed(P)
    xyz = (torch.rand(P, 3, generator=g) - 0.5) * 2
    vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    out = amd.ops.pe_points(xyz.to(dev), vd.to(dev)).cpu()
    want = torch.cat([O.positional_encoding(xyz, 10), torch.zeros(P, 1), O.positional_encoding(vd, 4), torch.zeros(P, 5)], dim=1)
    assert out.shape == (P, 96)
    assert float((out - want).abs().max()) < 5e-7            # (the kernels' sin / cos: 9e-8 of float64, the CPU's likewise)
    assert torch.equal(out[:, :3], xyz) and torch.equal(out[:, 64:67], vd) and float(out[:, 63].abs().max()) == 0.0 and float(out[:, 91:].abs().max()) == 0.0


def test_split_forward_saturates_beyond_the_fp16_range(amd, dev, oracle_params):
    """The split kernels' forward chain carries every operand as two fp16 pieces (22 bits; the backward chain keeps bf16 pieces for the
    gradients' range).  Activations are clamped to +-65504 in the ReLU's v_med3 and weights at packing time, so a decoder whose
    activations leave the fp16 range SATURATES -- finite outputs, not infinities or NaNs -- while activations of ordinary size (here up to
    a few hundred) are carried to fp32-accumulation accuracy.  The exact-fp32 kernels are the reference in both cases."""
    g = torch.Generator().manual_seed(3)
    P = 4096
    xyz = (torch.rand(P, 3, generator=g) - 0.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    lat = (torch.rand(1, 4, 256, generator=g) * 0.3).to(dev)
    for scale, expect_close in ((100.0, True), (3.0e5, False)):
        params = {k: v.clone() for k, v in oracle_params.items()}
        params["encoding_xyz.0.weight"] *= scale                       # first-layer activations of order scale
        params["shape_layer_1.0.weight"] /= scale                      # ... brought back by the next layer
        m = amd.CodeNeRF(3, 1); m.load_state_dict(params); m = m.to(dev)
        pk = m.packed_weights()
        out = {prec: amd.ops.decoder_fwd(xyz, vd, lat, pk, 3, 1, precision=prec)[:2] for prec in ("fp32", "bf16x3")}
        for t in out["bf16x3"]:
            assert bool(torch.isfinite(t).all())
        err = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(out["bf16x3"], out["fp32"]))
        print(f"[split forward, first-layer activations x {scale:g}] max relative difference from the fp32 kernels {err:.2e}")
        if expect_close:
            assert err < 2e-5, err


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_relu_bit_of_an_exactly_zero_preactivation_is_inactive(amd, dev, oracle_params, precision):
    """torch.relu has derivative 0 at 0 (the reference's autograd, src/model_supnerf.py:251-263).  Hidden units whose pre-activation is
    EXACTLY +0.0 for every point -- zero weight row, zero bias: dead or pruned units -- must be saved as inactive, and no gradient may flow
    through them (round 3's split forward took the sign bit of the pre-activation and called +0.0 active)."""
    from relu_bits import decode_relu_bits
    params = {k: v.clone() for k, v in oracle_params.items()}
    dead = [3, 64, 255]
    for name in ("shape_layer_2.0", "texture_layer_1.0", "encoding_xyz.0"):
        params[name + ".weight"][dead] = 0.0
        params[name + ".bias"][dead] = 0.0
    params["rgb.0.weight"][[5, 77]] = 0.0
    params["rgb.0.bias"][[5, 77]] = 0.0
    m = amd.CodeNeRF(3, 1); m.load_state_dict(params); m = m.to(dev)
    g = torch.Generator().manual_seed(2)
    P = 256
    xyz = (torch.rand(P, 3, generator=g) - 0.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    lat = torch.zeros(1, 4, 256, device=dev)                  # (no latent term: the dead units' inputs stay exactly zero)
    sig, rgb, masks = amd.ops.decoder_fwd(xyz, vd, lat, m.packed_weights(), 3, 1, save_masks=True, precision=precision)
    layers = decode_relu_bits(masks, P, 3, 1)                 # enc_xyz, shape 1..3, enc_viewdir, texture 1, rgb.0
    assert not bool(layers[0][:, dead].any()) and not bool(layers[2][:, dead].any()) and not bool(layers[5][:, dead].any())
    assert not bool(layers[6][:, [5, 77]].any())
    assert bool(layers[0].any()) and bool(layers[2].any())    # (the live units are a mix)
    # the backward applies the saved pattern: with the bits above, the kernel's d_xyz is the oracle's autograd on the SAME piecewise-linear
    # function (every ReLU differentiated with the saved 0/1 pattern; latent terms zero so that the dead units' inputs stay exactly zero)
    d_lat, d_xyz, d_dir = amd.ops.decoder_bwd(xyz, vd, lat, m.packed_weights(), masks, sig, torch.ones_like(sig), torch.ones_like(rgb), 3, 1,
                                              precision=precision)
    from oracle import supnerf_oracle as O
    xyz_c = xyz.cpu().requires_grad_()
    pc = {k: v for k, v in params.items()}
    sc = torch.zeros(1, 256); tc = torch.zeros(1, 256)
    with O.given_relu_masks(layers):
        for j in (1, 2, 3):               # (the oracle takes codes: zero latent layers give the zero latent terms the kernel was given)
            pc[f"shape_latent_layer_{j}.0.weight"] = torch.zeros_like(pc[f"shape_latent_layer_{j}.0.weight"]); pc[f"shape_latent_layer_{j}.0.bias"] = torch.zeros(256)
        pc["texture_latent_layer_1.0.weight"] = torch.zeros_like(pc["texture_latent_layer_1.0.weight"]); pc["texture_latent_layer_1.0.bias"] = torch.zeros(256)
        s_ref, c_ref = O.decoder_forward(pc, xyz_c[:, None, :], vd.cpu()[:, None, :], sc, tc)
        (s_ref.sum() + c_ref.sum()).backward()
    ref = xyz_c.grad
    err = float((d_xyz.cpu() - ref).abs().max() / ref.abs().max())
    assert err < (2e-4 if precision == "fp32" else 5e-4), err


@pytest.mark.parametrize("P,B", [(4096, 2), (1984, 1), (96, 3)])
def test_fp32_backward_kernels_agree(amd, dev, packed, P, B):
    """The exact-fp32 backward has two kernels: two waves per SIMD on 16x16x4 tiles (snr_mlp16_bwd.hip: the optimiser's backward) and one wave
    per SIMD on 32x32x2 tiles (snr_mlp_bwd.hip: taken when the training dumps are asked for).  Same products, same saved ReLU bits; the sums
    differ in association only (16- against 32-point partial rows, four against two lane groups in the encoding gradient)."""
    pk, _ = packed
    ops = amd.ops
    g = torch.Generator().manual_seed(P)
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev); vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    lat = (torch.randn(B, 4, 256, generator=g) * 0.3).to(dev)
    sig, rgb, masks = ops.decoder_fwd(xyz, vd, lat, pk, 3, 1, save_masks=True, precision="fp32")
    d_sig = torch.randn(P, generator=g).to(dev); d_rgb = torch.randn(P, 3, generator=g).to(dev)
    new = ops.decoder_bwd(xyz, vd, lat, pk, masks, sig, d_sig, d_rgb, 3, 1, precision="fp32")
    dumps = torch.empty(3 + 1 + 4, P, 256, device=dev)          # [layer][P][256], as DecoderTrain.backward allocates them
    old = ops.decoder_bwd(xyz, vd, lat, pk, masks, sig, d_sig, d_rgb, 3, 1, precision="fp32", layer_grads=dumps)
    for a, b, name in zip(new, old, ("d_latent", "d_xyz", "d_viewdir")):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * scale, name
