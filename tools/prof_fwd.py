"""Profiling target: a few launches of the fused forward/backward at the benchmark size (run under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, synthetic as SY, utils as U

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bwd = len(sys.argv) > 3 and sys.argv[3] == "bwd"      # "bwd": the forward that saves the ReLU bits + the backward
box = len(sys.argv) > 4 and sys.argv[4] == "box"      # "box": family B (NeRFRenderer.render_rays): box bounds, per-ray depths and Philox jitter in the prologue
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev); model.precision = prec
N, S = 4096, 64
ob = SY.synthetic_object(100)
g = torch.Generator().manual_seed(100)
sc = (torch.randn(1, 256, generator=g) * 0.3).to(dev); tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev)
with torch.no_grad():
    ro, vd = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
    near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S, generator=g))
    lat = model.latent_terms(sc, tc)
pk = model.packed_weights()
div = torch.full((1,), float(ob["obj_diag"]), device=dev)
cfg = ops.RenderCfg(S, ops.Z_SHARED, N, 3, 1, frame=U._frame(False, False, True), precision=prec)
cfg.latent_bias = model.latent_biases(lat)          # as model.fused_render passes them
if box:
    from supnerf_amd import renderer as R
    _, half, zs = R._box_constants(ob["wlh"], 1, dev)
    cfg = ops.RenderCfg(S, ops.Z_BOX, N, 3, 1, white_bkgd=True, metric_z=True, precision=prec, box_half=half)
    cfg.latent_bias = model.latent_biases(lat)
    z, div = None, None
    ro = ro.contiguous()
    if not bwd:
        for _ in range(n):
            ops.render_fwd(ro, vd, None, None, zs, lat, pk, cfg)
    else:
        latg = lat.clone().requires_grad_(); rog = ro.clone().requires_grad_(); vdg = vd.clone().requires_grad_()
        for _ in range(n):
            rgb, depth, acc = ops.FusedRender.apply(rog, vdg, None, None, zs, latg, pk, cfg)
            (rgb.sum() + acc.sum() + depth.sum()).backward()
elif not bwd:
    for _ in range(n):
        ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)
else:
    latg = lat.clone().requires_grad_(); rog = ro.clone().requires_grad_(); vdg = vd.clone().requires_grad_()
    for _ in range(n):
        rgb, depth, acc = ops.FusedRender.apply(rog, vdg, z, div, None, latg, pk, cfg)
        (rgb.sum() + acc.sum()).backward()
torch.cuda.synchronize()
print("done", prec, n)
