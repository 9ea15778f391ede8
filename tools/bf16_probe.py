"""Development probe: split-bf16 kernels vs the exact fp32 kernels at the benchmark size (parity + timing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, synthetic as SY, utils as U

dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev)
N, S = 4096, 64
ob = SY.synthetic_object(100)
img, mask = SY.synthetic_targets(100, 64)
g = torch.Generator().manual_seed(100)
sc = (torch.randn(1, 256, generator=g) * 0.3).to(dev); tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev)
jit = torch.rand(S, generator=g)
with torch.no_grad():
    ro, vd = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
    near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z = U._shared_depths(near, far, S, dev, jitter=jit)
    lat = model.latent_terms(sc, tc)
pk = model.packed_weights()
div = torch.full((1,), float(ob["obj_diag"]), device=dev)
out = {}
for prec in ("fp32", "bf16x3"):
    cfg = ops.RenderCfg(S, ops.Z_SHARED, N, 3, 1, frame=U._frame(False, False, True), precision=prec)
    o = ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg, save_for_bwd=True)
    torch.cuda.synchronize()
    out[prec] = o
    for _ in range(5):
        ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{prec}: fwd {ms:.3f} ms  {N/ms*1e3/1e6:.2f} Mrays/s  {N*57.56e6/ms/1e9:.1f} TFLOP/s algorithmic", flush=True)
a, b = out["fp32"], out["bf16x3"]
tgt = img.reshape(-1, 3).to(dev); occ = mask.reshape(-1, 1).to(dev); fg = occ.clamp_min(0)
ps = lambda rgb: float(-10 * torch.log10(((rgb - tgt) ** 2 * fg).sum() / (fg.sum() + 1e-9)))
print(f"bf16x3 vs fp32: rgb max {float((a[0]-b[0]).abs().max()):.2e}  depth mean {float((a[1]-b[1]).abs().mean()):.2e} max {float((a[1]-b[1]).abs().max()):.2e}  "
      f"acc max {float((a[2]-b[2]).abs().max()):.2e}  PSNR delta {abs(ps(a[0])-ps(b[0])):.2e} dB  sigma max rel {float(((a[3]-b[3]).abs()/(a[3].abs()+1e-6)).max()):.2e}  "
      f"mask bits differing {int((a[5]!=b[5]).sum())} bytes of {a[5].numel()}")
# backward
for prec in ("fp32", "bf16x3"):
    model.precision = prec
    scg, tcg = sc.clone().requires_grad_(), tc.clone().requires_grad_()
    pose = ob["cam_pose"].to(dev).requires_grad_()
    def it():
        r, v = U.get_rays(ob["K"], pose, ob["roi"], uv_steps=[64, 64])
        cfg = ops.RenderCfg(S, ops.Z_SHARED, N, 3, 1, frame=U._frame(False, False, True), precision=prec)
        rgb, depth, acc = model.fused_render(r, v, z, div, None, scg, tcg, cfg)
        loss = ((rgb - tgt) ** 2).mean() + 0.1 * acc.mean()
        scg.grad = tcg.grad = pose.grad = None
        loss.backward()
    it(); torch.cuda.synchronize()
    out[prec + "_g"] = (scg.grad.clone(), tcg.grad.clone(), pose.grad.clone())
    for _ in range(3): it()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): it()
    e1.record(); torch.cuda.synchronize()
    print(f"{prec}: fwd+bwd iteration {e0.elapsed_time(e1)/20:.3f} ms", flush=True)
ga, gb = out["fp32_g"], out["bf16x3_g"]
for n, x, y in zip(("d_shape", "d_texture", "d_pose"), ga, gb):
    print(f"  {n}: rel max err {float((x-y).abs().max()/x.abs().max()):.2e}")
