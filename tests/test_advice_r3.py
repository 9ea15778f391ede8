"""Regression tests for the round-3 advisor findings: each is a silent-wrong-answer class, so each gets a test that fails on
the round-3 tree."""
import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O

pytestmark = pytest.mark.gpu


def test_training_pack_cache_cannot_serve_a_freed_models_weights(oracle_params):
    """ops._packed_for was keyed by (data_ptr, _version, device) only: model A trains a step, is freed, the caching allocator hands its
    addresses to model B built the same way (same version counters) -- and B ran on A's packed weights.  The entry now proves identity."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    xyz = (torch.rand(64, 32, 3, generator=g) - 0.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(64, 32, 3, generator=g), dim=-1).to(dev)
    sc, tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev), (torch.randn(1, 256, generator=g) * 0.3).to(dev)

    def build(scale):
        params = {k: v.clone() for k, v in oracle_params.items()}
        params["rgb.2.weight"] = params["rgb.2.weight"] * scale          # a different function, same construction sequence
        m = A.CodeNeRF(3, 1)
        m.load_state_dict(params)
        m = m.to(dev)
        m.train_decoder_weights = True
        m.precision = "fp32"
        return m, params

    ptrs = []
    for scale in (1.0, 3.0, 0.5):
        torch.cuda.empty_cache()
        m, params = build(scale)
        ptrs.append(m.rgb[2].weight.data_ptr())
        sig, rgb = m(xyz, vd, sc, tc)
        s_ref, c_ref = O.decoder_forward(params, xyz.cpu(), vd.cpu(), sc.cpu(), tc.cpu())
        assert float((rgb.detach().cpu() - c_ref).abs().max()) < 2e-5, scale
        rgb.sum().backward()
        del m, sig, rgb
    # (informative: whether the allocator really re-used the addresses on this run; the assertion above must hold either way)
    print("[pack cache] weight addresses of the three models:", [hex(p) for p in ptrs])


def test_4x4_pose_on_the_gpu_takes_the_references_bounds(oracle_params, golden):
    """A homogeneous (4,4) pose: the reference's sphere bounds take the norm of the WHOLE last column (src/utils.py:468,
    ``cam_pose[:, -1]``), including the 1.  The fused pose -> rays launch computes |t| of three entries, so a (4,4) pose must take the
    torch formulation on the GPU as it does on the CPU: same depths either way."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    m = A.CodeNeRF(3, 1); m.load_state_dict(oracle_params); m = m.to(dev); m.precision = "fp32"
    g = golden("render_a_nusc")
    pose44 = torch.cat([g["cam_pose"], torch.tensor([[0.0, 0.0, 0.0, 1.0]])], dim=0)
    outs = []
    for pose in (pose44, pose44.to(dev)):
        with A.utils.jitter_override(g["jitter"]), torch.no_grad():
            outs.append(A.utils.render_rays_v2(m, dev, g["img"], g["mask_occ"], pose, np.float32(g["obj_diag"]), g["K"], g["roi"], int(g["n_samples"]),
                                               g["shapecode"].to(dev), g["texturecode"].to(dev), int(g["shapenet_obj_cood"]), 0, im_sz=int(g["im_sz"])))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert float((a - b).abs().max()) < 2e-5
    # and the oracle on the same 4x4 pose (its sphere bounds restate the reference's full-column norm)
    ref = O.render_rays_v2(oracle_params, g["img"], g["mask_occ"], pose44, float(g["obj_diag"]), g["K"], g["roi"], int(g["n_samples"]), g["shapecode"],
                           g["texturecode"], bool(g["shapenet_obj_cood"]), im_sz=int(g["im_sz"]), jitter=g["jitter"])
    assert float((outs[1][1].cpu() - ref[1]).abs().max()) < 2e-4 and float((outs[1][0].cpu() - ref[0]).abs().max()) < 5e-5


def test_box_sampling_with_padded_objects_keeps_torch_rand_likes_stream(oracle_params):
    """Family B with in-kernel jitter, S = 16 and an odd ray count per object + the codes' gradient wanted: FusedRender pads every object
    to whole 32-point tiles, which shifts the kernel's Philox indices -- the reference draw (``torch.rand_like`` of the caller's (N,S)
    table) is now materialised before padding: same rendered values as with that table injected, same generator consumption."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    m = A.CodeNeRF(3, 1); m.load_state_dict(oracle_params); m = m.to(dev); m.precision = "fp32"
    ob = O.synthetic_object(7)
    S, n = 16, 37                                          # 37 x 16 = 592 points: not a multiple of 32
    rays_o, viewdir = O.pixel_rays(ob["K"], ob["cam_pose"], ob["roi"], uv_steps=[8, 8])
    rays_o, viewdir = rays_o[:n].to(dev).contiguous(), viewdir[:n].to(dev).contiguous()
    g = torch.Generator().manual_seed(3)
    sc0, tc0 = torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3
    from supnerf_amd import renderer as R
    diag, half, zs = R._box_constants(ob["wlh"], 1, dev)

    def run(jitter):
        sc, tc = sc0.to(dev).requires_grad_(), tc0.to(dev).requires_grad_()
        cfg = A.ops.RenderCfg(S, A.ops.Z_BOX, n, 3, 1, white_bkgd=True, metric_z=True, precision="fp32", box_half=half)
        out = m.fused_render(rays_o / 1.0, viewdir, jitter, None, zs, sc, tc, cfg)
        (out[0].sum() + out[2].sum()).backward()
        return [t.detach() for t in out], sc.grad.clone()

    torch.manual_seed(77)
    junk = torch.rand(100, device=dev)
    jit = torch.rand_like(torch.empty(n, S, device=dev))
    after = torch.rand(4, device=dev)
    want, g_want = run(jit)
    torch.manual_seed(77)
    junk2 = torch.rand(100, device=dev)
    got, g_got = run(None)
    after2 = torch.rand(4, device=dev)
    assert torch.equal(junk, junk2) and torch.equal(after, after2), "the generator did not advance like torch.rand_like"
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert torch.equal(g_got, g_want)
