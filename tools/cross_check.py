"""Randomised cross-check of the two arithmetic modes: fused render forward + backward, fp32 kernels vs split-bf16 kernels,
over random shapes (objects, rays per object, samples per ray, depth modes).  Prints the worst relative differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, synthetic as SY, utils as U

dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev)
gen = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {}
n_cases = 0
for case in range(40):
    S = [4, 8, 16, 32, 64, 128][int(torch.randint(0, 6, (1,), generator=gen))]
    B = int(torch.randint(1, 5, (1,), generator=gen))
    per = int(torch.randint(1, 40, (1,), generator=gen)) * max(1, 32 // S) * 2     # rays per object: whole 32-point tiles, even
    z_mode = [ops.Z_SHARED, ops.Z_PER_OBJECT, ops.Z_PER_RAY][int(torch.randint(0, 3, (1,), generator=gen))]
    white = bool(torch.randint(0, 2, (1,), generator=gen))
    metric = z_mode == ops.Z_PER_RAY and bool(torch.randint(0, 2, (1,), generator=gen))
    N = B * per
    ro = (torch.randn(N, 3, generator=gen) * 0.05 + torch.tensor([0.0, -2.2, 0.2])).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen) * 0.1 + torch.tensor([0.0, 1.0, 0.0]), dim=-1).to(dev)
    shape = {ops.Z_SHARED: (S,), ops.Z_PER_OBJECT: (B, S), ops.Z_PER_RAY: (N, S)}[z_mode]
    t = (torch.sort(torch.rand(*shape, generator=gen), dim=-1)[0] * 1.5 + 1.4).to(dev)
    sc = (torch.randn(B, 256, generator=gen) * 0.3).to(dev); tc = (torch.randn(B, 256, generator=gen) * 0.3).to(dev)
    div = (torch.rand(B, generator=gen) * 0.5 + 0.8).to(dev)
    zs = (torch.rand(B, generator=gen) + 2.0).to(dev) if metric else None
    wts = [torch.randn(N, 3, generator=gen).to(dev), (torch.randn(N, generator=gen) * 0.1).to(dev), torch.randn(N, generator=gen).to(dev)]
    res = {}
    for prec in ("fp32", "bf16x3"):
        leaves = [x.clone().requires_grad_() for x in (ro, vd, sc, tc)] + ([t.clone().requires_grad_()] if z_mode == ops.Z_PER_RAY else [])
        tt = leaves[4] if z_mode == ops.Z_PER_RAY else t
        cfg = ops.RenderCfg(S, z_mode, per, 3, 1, frame=U._frame(False, False, True), white_bkgd=white, metric_z=metric, precision=prec)
        out = model.fused_render(leaves[0], leaves[1], tt, div, zs, leaves[2], leaves[3], cfg)
        sum((a * b).sum() for a, b in zip(out, wts)).backward()
        res[prec] = [o.detach() for o in out] + [l.grad for l in leaves]
    # mask-matched: ONE exact-fp32 forward saves sigmas / colours / ReLU bits, both backward kernels differentiate that same piecewise-linear
    # function -- what is left between them is arithmetic, not ReLU flips (the comparison above lets each arithmetic save its own bits)
    with torch.no_grad():
        lat = model.latent_terms(sc, tc).contiguous()
        pk = model.packed_weights()
        cfg32 = ops.RenderCfg(S, z_mode, per, 3, 1, frame=U._frame(False, False, True), white_bkgd=white, metric_z=metric, precision="fp32")
        _, _, _, sig, rgbs, masks = ops.render_fwd(ro, vd, t, div, zs, lat, pk, cfg32, save_for_bwd=True)
        mm = {}
        for prec in ("fp32", "bf16x3"):
            cfg_b = ops.RenderCfg(S, z_mode, per, 3, 1, frame=U._frame(False, False, True), white_bkgd=white, metric_z=metric, precision=prec)
            mm[prec] = ops.render_bwd(ro, vd, t, div, zs, lat, pk, cfg_b, sig, rgbs, masks, wts[0], wts[1], wts[2], need_t=(z_mode == ops.Z_PER_RAY))
        for nme, a, b in zip(("mm d_rays_o", "mm d_rays_d", "mm d_t", "mm d_latent"), mm["fp32"], mm["bf16x3"]):
            if a is None: continue
            rel = float((a - b).abs().max() / (a.abs().max() + 1e-12))
            if not torch.isfinite(b).all(): rel = float("inf")
            if rel > worst.get(nme, (0,))[0]:
                worst[nme] = (rel, dict(S=S, B=B, per=per, z_mode=z_mode, white=white, metric=metric))
    names = ["rgb", "depth", "acc", "d_rays_o", "d_rays_d", "d_shape", "d_texture", "d_t"]
    for nme, a, b in zip(names, res["fp32"], res["bf16x3"]):
        rel = float((a - b).abs().max() / (a.abs().max() + 1e-12))
        if not torch.isfinite(b).all():
            rel = float("inf")
        if rel > worst.get(nme, (0,))[0]:
            worst[nme] = (rel, dict(S=S, B=B, per=per, z_mode=z_mode, white=white, metric=metric))
    n_cases += 1
print(n_cases, "cases; worst relative difference bf16x3 vs fp32 (max-norm):")
for k, v in worst.items():
    print(f"  {k:12s} {v[0]:.2e}   at {v[1]}")
