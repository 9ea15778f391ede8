"""Cost of the family-B prologue (SNR_Z_BOX) inside the fused forward / backward kernels against the family-A launch of the same size:
box bounds + depths with a jitter TABLE, and with the in-kernel Philox jitter.  Device events, 4096 rays x 64 samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, utils as U, renderer as R, synthetic as O

dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params(seed=0)); model = model.to(dev)
ob = O.synthetic_object(100)
N, S = 4096, 64
with torch.no_grad():
    ro, vd = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
    ro = ro.contiguous()
    near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S))
    g = torch.Generator().manual_seed(0)
    lat = model.latent_terms((torch.randn(1, 256, generator=g) * 0.3).to(dev), (torch.randn(1, 256, generator=g) * 0.3).to(dev))
packed = model.packed_weights()
div = torch.full((1,), float(ob["obj_diag"]), device=dev)
_, half, zs = R._box_constants(ob["wlh"], 1, dev)
jit = torch.rand(N, S, device=dev)


def ev(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for p in ("fp32", "bf16x3"):
    cfg_a = ops.RenderCfg(S, ops.Z_SHARED, N, 3, 1, frame=U._frame(False, False, True), precision=p)
    cfg_b = ops.RenderCfg(S, ops.Z_BOX, N, 3, 1, white_bkgd=True, metric_z=True, precision=p, box_half=half)
    cfg_a.latent_bias = cfg_b.latent_bias = model.latent_biases(lat)
    for save in (False, True):
        a = ev(lambda: ops.render_fwd(ro, vd, z, div, None, lat, packed, cfg_a, save_for_bwd=save))
        b_tab = ev(lambda: ops.render_fwd(ro, vd, jit, None, zs, lat, packed, cfg_b, save_for_bwd=save))
        b_rng = ev(lambda: ops.render_fwd(ro, vd, None, None, zs, lat, packed, cfg_b, save_for_bwd=save))
        print(f"{p} forward{' + ReLU bits' if save else ''}: family A {a:.4f} ms, box + jitter table {b_tab:.4f} ms, box + Philox {b_rng:.4f} ms")
    fa = ops.render_fwd(ro, vd, z, div, None, lat, packed, cfg_a, save_for_bwd=True)
    fb = ops.render_fwd(ro, vd, None, None, zs, lat, packed, cfg_b, save_for_bwd=True)
    fc = ops.render_fwd(ro, vd, jit, None, zs, lat, packed, cfg_b, save_for_bwd=True)
    d = [torch.rand(N, 3, device=dev), torch.rand(N, device=dev), torch.rand(N, device=dev)]
    a = ev(lambda: ops.render_bwd(ro, vd, z, div, None, lat, packed, cfg_a, fa[3], fa[4], fa[5], *d), 30)
    b_tab = ev(lambda: ops.render_bwd(ro, vd, jit, None, zs, lat, packed, cfg_b, fc[3], fc[4], fc[5], *d), 30)
    b_rng = ev(lambda: ops.render_bwd(ro, vd, None, None, zs, lat, packed, cfg_b, fb[3], fb[4], fb[5], *d), 30)
    print(f"{p} backward: family A {a:.4f} ms, box + jitter table {b_tab:.4f} ms, box + Philox {b_rng:.4f} ms")
