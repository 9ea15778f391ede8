"""Kernel time against the number of workgroup rounds: the fused fp32 render forward / backward at N rays x 64 samples for N = 512 .. 8192
(one 64-point workgroup per ray, 512 workgroup slots on the chip -> N / 512 rounds).  A linear fit separates the per-round time from the
fixed cost of a launch (ramp, tail, the latent reduction).   usage: python tools/_diag/grid_scan.py [precision 0|1]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import supnerf_amd as A
from supnerf_amd import _lib, ops, synthetic as SY, utils as U
lib = _lib.lib(); dev = torch.device("cuda:0")
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
model = A.CodeNeRF(3, 1); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev)
S = 64
ob = SY.synthetic_object(100)
g = torch.Generator().manual_seed(100)
sc = (torch.randn(1, 256, generator=g) * 0.3).to(dev); tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev)
with torch.no_grad():
    ro0, vd0 = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
    near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S, generator=g))
    lat = model.latent_terms(sc, tc).contiguous()
pk = model.packed_weights(); lb = model.latent_biases(lat)
div = torch.full((1,), float(ob["obj_diag"]), device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, n=20):
    for _ in range(4): fn()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n)
    return best
rows = []
for N in (512, 1024, 2048, 4096, 8192):
    reps = (N + 4095) // 4096
    ro = ro0.repeat(reps, 1)[:N].contiguous(); vd = vd0.repeat(reps, 1)[:N].contiguous()
    a = ops._render_args(ro, vd, z, div, None, lat, pk, U._frame(False, False, True), 1.0, ops.Z_SHARED, 0, N, S, 3, 1, prec, latent_bias=lb)
    rgb = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev); acc = torch.empty(N, device=dev)
    sig = torch.empty(N * S, device=dev); rgbs = torch.empty(N * S, 3, device=dev)
    masks = torch.empty(int(lib.snr_mask_bytes(N * S, 3, 1)), dtype=torch.uint8, device=dev)
    d_rgb = torch.rand(N, 3, device=dev); d_depth = torch.rand(N, device=dev); d_acc = torch.rand(N, device=dev)
    d_lat = torch.empty_like(lat); d_o = torch.zeros(N, 3, device=dev); d_d = torch.zeros(N, 3, device=dev)
    wsb = lib.snr_render_bwd_ws_bytes(C.byref(a)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    f0 = lambda: lib.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), None, None, None, st())
    f1 = lambda: lib.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), st())
    b = lambda: lib.snr_render_bwd(C.byref(a), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), d_rgb.data_ptr(), d_depth.data_ptr(), d_acc.data_ptr(),
                                   d_lat.data_ptr(), d_o.data_ptr(), d_d.data_ptr(), None, ws.data_ptr(), wsb, st())
    bn = lambda: lib.snr_render_bwd(C.byref(a), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), d_rgb.data_ptr(), d_depth.data_ptr(), d_acc.data_ptr(),
                                    None, d_o.data_ptr(), d_d.data_ptr(), None, ws.data_ptr(), wsb, st())
    assert f1() == 0 and b() == 0
    rows.append((N, timed(f0), timed(f1), timed(b), timed(bn)))
    print(f"N {N:5d} rounds {N / 512:4.1f}  fwd {rows[-1][1]:.4f}  fwd+bits {rows[-1][2]:.4f}  bwd {rows[-1][3]:.4f}  bwd without latent gradient {rows[-1][4]:.4f} ms", flush=True)
import numpy as np
x = np.array([r[0] / 512 for r in rows])
for k, name in ((1, "fwd"), (2, "fwd+bits"), (3, "bwd"), (4, "bwd no-latent")):
    y = np.array([r[k] for r in rows]); sl, ic = np.polyfit(x, y, 1)
    print(f"{name:14s} per round {sl * 1e3:.1f} us, fixed {ic * 1e3:.1f} us   (ideal per round at 2.4 GHz: {512 * 64 * 450560 / (1024 * 32 * 2.4e9) * 1e6:.1f} us)")
