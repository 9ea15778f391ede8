"""Diagnostic: effective shader clock under the exact-fp32 fused forward (library built by `tools/build_diag.sh clk` with -DSNR_STAMPS:
every workgroup records s_memtime / s_memrealtime at both ends).  clock = d(memtime) / d(memrealtime) x 100 MHz, after >= 2 s of
back-to-back launches (MI355X_MICROARCH.md, in-kernel clock check).  usage: python tools/clock32.py"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import supnerf_amd as A
from supnerf_amd import _lib, ops, synthetic as SY, utils as U
good = _lib.lib()
stamps = C.CDLL(os.path.join(ROOT, "tools", "_diag", "libsupnerf_stamps_clk.so"), mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
stamps.snr_render_fwd.restype = C.c_int
stamps.snr_render_fwd.argtypes = _lib._SIGS["snr_render_fwd"][1]
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev)
N, S = 4096, 64
ob = SY.synthetic_object(100)
g = torch.Generator().manual_seed(100)
sc = (torch.randn(1, 256, generator=g) * 0.3).to(dev); tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev)
with torch.no_grad():
    ro, vd = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
    near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S, generator=g))
    lat = model.latent_terms(sc, tc).contiguous()
pk = model.packed_weights()
div = torch.full((1,), float(ob["obj_diag"]), device=dev)
a = ops._render_args(ro.contiguous(), vd.contiguous(), z, div, None, lat, pk, U._frame(False, False, True), 1.0, ops.Z_SHARED, 0, N, S, 3, 1, 0)
rgb = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev); acc = torch.empty(N, device=dev)
dbg = torch.zeros(N * S, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
t0 = time.time(); n = 0
while time.time() - t0 < 2.5:
    for _ in range(50):
        assert stamps.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), dbg.data_ptr(), None, None, st()) == 0
    torch.cuda.synchronize(); n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    stamps.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), dbg.data_ptr(), None, None, st())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
t = dbg.cpu().numpy().view(np.uint64).reshape(-1, 16)[: N * S // 128].astype(np.int64)
cyc = t[:, 2] - t[:, 0]; real = t[:, 3] - t[:, 1]
clk = cyc / np.maximum(real, 1) * 100.0        # MHz
span_real = (t[:, 3].max() - t[:, 1].min()) / 100.0   # us
print(f"launches warmed: {n}; kernel {ms:.4f} ms by events, {span_real/1000:.4f} ms first-start..last-end by s_memrealtime")
print(f"per-workgroup shader cycles: median {np.median(cyc):.0f}  (min {cyc.min()}, max {cyc.max()}); duration median {np.median(real)/100:.1f} us")
print(f"effective clock: median {np.median(clk):.0f} MHz (p10 {np.percentile(clk,10):.0f}, p90 {np.percentile(clk,90):.0f})")
mf = 1024 * 8 + 256 + 512 + 128      # MFMAs per wave: 6 layers x 8 chunks x 128 + view chunk 128 ... printed for reference only
flops_wg = 2.0 * 128 * (64 * 256 + 5 * 256 * 256 + 288 * 256 + 256 * 128)
peak_at_clk = 256 * 4 * (32 * 32 * 2 * 2 / 64.0) * np.median(clk) * 1e6
print(f"MFMA flops per workgroup {flops_wg:.3e}; fp32 MFMA peak at the measured clock {peak_at_clk/1e12:.1f} TFLOP/s; "
      f"achieved {flops_wg * (N * S // 128) / (ms * 1e-3) / 1e12:.1f} TFLOP/s = {flops_wg * (N * S // 128) / (ms * 1e-3) / peak_at_clk:.3f} of it")

names = ["prologue", "enc_xyz + boundary"] + [f"layer {i} + boundary" for i in range(1, 7)] + ["(to rgb.0)", "rgb.0 chunks"]
ph = t[:, 4:14] - t[:, :1]
med = np.median(ph, axis=0)
floor = [0, 2 * 128 * 64] + [8 * 128 * 64] * 4 + [9 * 128 * 64] + [8 * 128 * 64] + [0, 8 * 64 * 64]
print("phase, cycles from workgroup start (median), this phase, its MFMA cycles")
prev = 0
for n_, m_, f_ in zip(names, med, floor):
    print(f"  {n_:22s} {m_:9.0f}  +{m_ - prev:8.0f}   {f_:7d}")
    prev = m_
print(f"  {'heads, composite, end':22s} {np.median(cyc):9.0f}  +{np.median(cyc) - prev:8.0f}")
