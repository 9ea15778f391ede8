"""Regression test for round 2's GPU memory-access fault (``ops.render_bwd`` handed the C ABI the data_ptr of ``get_rays``' origins, a
stride-0 ``expand`` of the pose's translation).  The C ABI takes dense buffers and cannot know strides, so every operator wrapper must make
its operands dense -- and the one pointer helper every wrapper uses refuses a strided tensor outright.  Each direct-call operator gets
strided views of every operand it takes (stride-0 expand, transposed, column slice of a wider buffer) and must return what it returns
for dense copies."""
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


def strided(t):
    """The same values behind non-trivial strides: a column slice of a wider buffer (2-D+) or every second element of a longer one (1-D)."""
    if t.dim() == 1:
        big = torch.zeros(t.shape[0] * 2, device=t.device, dtype=t.dtype)
        big[::2] = t
        v = big[::2]
    else:
        big = torch.zeros(*t.shape[:-1], t.shape[-1] + 5, device=t.device, dtype=t.dtype)
        big[..., 2:2 + t.shape[-1]] = t
        v = big[..., 2:2 + t.shape[-1]]
    assert not v.is_contiguous() and torch.equal(v, t)
    return v


def test_pointer_helper_refuses_strided_and_cpu_tensors(amd, dev):
    ops = amd.ops
    t = torch.rand(8, 3, device=dev)
    ops._p(t)
    with pytest.raises(amd.SnrError):
        ops._p(t.t())
    with pytest.raises(amd.SnrError):
        ops._p(t[:, :2])
    with pytest.raises(amd.SnrError):
        ops._p(torch.rand(3))
    with pytest.raises(amd.SnrError):
        ops._ptr(t.double())


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_render_bwd_takes_strided_views(amd, dev, oracle_params, precision):
    ops, U = amd.ops, amd.utils
    model = amd.CodeNeRF(3, 1)
    model.load_state_dict(oracle_params)
    model = model.to(dev)
    ob = O.synthetic_object(5)
    N, S = 256, 64
    pose = ob["cam_pose"].to(dev)
    with torch.no_grad():
        _, viewdir = U.get_rays(ob["K"], pose, ob["roi"], uv_steps=[16, 16])
        rays_o = pose[:, 3].expand(N, 3)                      # THE operand of the round-2 fault: get_rays' origins used to be this stride-0 view
        assert rays_o.stride(0) == 0
        near, far = U._sphere_bounds(pose, ob["obj_diag"])
        z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S))
        g = torch.Generator().manual_seed(0)
        lat = model.latent_terms((torch.randn(1, 256, generator=g) * 0.3).to(dev), (torch.randn(1, 256, generator=g) * 0.3).to(dev))
    packed = model.packed_weights()
    div = torch.full((1,), float(ob["obj_diag"]), device=dev)
    cfg = ops.RenderCfg(S, ops.Z_SHARED, N, 3, 1, frame=U._frame(False, False, True), precision=precision)
    d_rgb, d_depth, d_acc = torch.rand(N, 3, device=dev), torch.rand(N, device=dev), torch.rand(N, device=dev)

    dense = [t.contiguous() for t in (rays_o, viewdir)]
    fw = ops.render_fwd(dense[0], dense[1], z, div, None, lat, packed, cfg, save_for_bwd=True)
    want = ops.render_bwd(dense[0], dense[1], z, div, None, lat, packed, cfg, fw[3], fw[4], fw[5], d_rgb, d_depth, d_acc)

    # every operand strided: forward and backward
    fw_s = ops.render_fwd(rays_o, viewdir.t().contiguous().t(), strided(z), div.expand(1), None, strided(lat), packed, cfg, save_for_bwd=True)
    for a, b in zip(fw[:5], fw_s[:5]):
        assert torch.equal(a, b)
    got = ops.render_bwd(rays_o, viewdir.t().contiguous().t(), strided(z), div.expand(1), None, strided(lat), packed, cfg,
                         strided(fw[3]), strided(fw[4]), fw[5], strided(d_rgb), strided(d_depth), strided(d_acc))
    for a, b in zip(want, got):
        assert (a is None and b is None) or torch.equal(a, b)
    # operand sizes are checked on the host: a short buffer must not reach the kernel
    with pytest.raises(amd.SnrError):
        ops.render_bwd(dense[0], dense[1], z, div, None, lat, packed, cfg, fw[3][:-64], fw[4], fw[5], d_rgb, d_depth, d_acc)
    with pytest.raises(amd.SnrError):
        ops.render_fwd(dense[0], dense[1], z[:-1], div, None, lat, packed, cfg)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_decoder_bwd_takes_strided_views(amd, dev, oracle_params, precision):
    ops = amd.ops
    model = amd.CodeNeRF(3, 1)
    model.load_state_dict(oracle_params)
    model = model.to(dev)
    packed = model.packed_weights()
    g = torch.Generator().manual_seed(1)
    P = 512
    xyz = (torch.rand(P, 3, generator=g) - 0.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    lat = torch.rand(2, 4, 256, generator=g).to(dev)
    sig, rgb, masks = ops.decoder_fwd(xyz, vd, lat, packed, 3, 1, save_masks=True, precision=precision)
    d_sig, d_rgb = torch.rand(P, device=dev), torch.rand(P, 3, device=dev)
    want = ops.decoder_bwd(xyz, vd, lat, packed, masks, sig, d_sig, d_rgb, 3, 1, precision=precision)
    sig_s, rgb_s, _ = ops.decoder_fwd(strided(xyz), vd.t().contiguous().t(), strided(lat), packed, 3, 1, save_masks=True, precision=precision)
    assert torch.equal(sig, sig_s) and torch.equal(rgb, rgb_s)
    got = ops.decoder_bwd(strided(xyz), vd.t().contiguous().t(), strided(lat), packed, masks, strided(sig), strided(d_sig), strided(d_rgb), 3, 1,
                          precision=precision)
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_composite_bwd_takes_strided_views(amd, dev):
    ops = amd.ops
    N, S = 96, 64
    sig, rgbs = torch.rand(N, S, device=dev), torch.rand(N, S, 3, device=dev)
    z = torch.sort(torch.rand(N, S, device=dev) * 3 + 1, dim=-1)[0]
    d = [torch.rand(N, 3, device=dev), torch.rand(N, device=dev), torch.rand(N, device=dev)]
    want = ops.composite_bwd(sig, rgbs, z, ops.Z_PER_RAY, True, 0, *d, True)
    got = ops.composite_bwd(strided(sig), strided(rgbs), z.t().contiguous().t(), ops.Z_PER_RAY, True, 0, *[strided(t) for t in d], True)
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    out = ops.composite_fwd(sig, rgbs, z, ops.Z_PER_RAY, True)
    out_s = ops.composite_fwd(strided(sig), strided(rgbs), z.t().contiguous().t(), ops.Z_PER_RAY, True)
    for a, b in zip(out, out_s):
        assert torch.equal(a, b)
