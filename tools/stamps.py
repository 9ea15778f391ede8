"""Diagnostic: per-phase timeline of the bf16x3 forward kernel from in-kernel s_memtime stamps
(library built with -DSNR_STAMPS into tools/_diag/; outputs of that build are not valid renders)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import supnerf_amd as A
from supnerf_amd import _lib, ops, synthetic as SY, utils as U
good = _lib.lib()                                   # regular library for packing
stamps = C.CDLL(os.path.join(ROOT, "tools", "_diag", os.environ.get("SNR_STAMP_LIB", "libsupnerf_stamps.so")), mode=os.RTLD_NOW | os.RTLD_DEEPBIND)   # own symbols first
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
ro, vd = ro.contiguous(), vd.contiguous()
PREC = int(os.environ.get("SNR_STAMP_PRECISION", "1"))        # 1 = bf16x3 (default), 0 = fp32 (backward timeline only: the fp32 forward's is tools/clock32.py)
a = ops._render_args(ro, vd, z, div, None, lat, pk, U._frame(False, False, True), 1.0, ops.Z_SHARED, 0, N, S, 3, 1, PREC,
                     latent_bias=model.latent_biases(lat))
rgb = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev); acc = torch.empty(N, device=dev)
dbg = torch.zeros(N * S, device=dev)               # "sigmas" buffer receives the stamps: 16 x u64 per wave tile
for _ in range(3):
    rc = stamps.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), dbg.data_ptr(), None, None,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
torch.cuda.synchronize()
t = dbg.cpu().numpy().view(np.uint64).reshape(-1, 16)[: N * S // 32].astype(np.int64)
names = ["start", "staged", "PE done", "L0 done"] + [f"layer {i} done" for i in range(1, 7)] + ["", "", "rgb0 done", "heads done", "end"]
d = t - t[:, :1]
print("cycles from kernel start (median over %d wave tiles):" % len(t))
prev = 0
for i, n in enumerate(names):
    if not n: continue
    m = float(np.median(d[:, i]))
    print(f"  {n:14s} {m:10.0f}  (+{m - prev:8.0f})")
    prev = m
print("first start -> last end (cycles):", int(t[:, 14].max() - t[:, 0].min()), " waves:", len(t))

# ---------------------------------------------------------------- forward that also saves the ReLU bits (the optimiser's forward)
masks_dbg = torch.empty(int(good.snr_mask_bytes(N * S, 3, 1)), dtype=torch.uint8, device=dev)
rgbs_dbg = torch.empty(N * S, 3, device=dev)
dbg.zero_()
for _ in range(3):
    rc = stamps.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), dbg.data_ptr(), rgbs_dbg.data_ptr(), masks_dbg.data_ptr(),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
torch.cuda.synchronize()
t2 = dbg.cpu().numpy().view(np.uint64).reshape(-1, 16)[: N * S // 32].astype(np.int64)
d2 = t2 - t2[:, :1]
print("forward WITH ReLU-bit capture:", "  ".join(f"{n or i}:{int(np.median(d2[:, i]))}" for i, n in enumerate(names) if n))

# ---------------------------------------------------------------- backward timeline (stamps land in the d_t buffer)
sig = torch.empty(N * S, device=dev); rgbs = torch.empty(N * S, 3, device=dev)
masks = torch.empty(int(good.snr_mask_bytes(N * S, 3, 1)), dtype=torch.uint8, device=dev)
rc = good.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(),
                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
assert rc == 0, rc
stamps.snr_render_bwd.restype = C.c_int
stamps.snr_render_bwd.argtypes = _lib._SIGS["snr_render_bwd"][1]
stamps.snr_render_bwd_ws_bytes.restype = C.c_size_t
stamps.snr_render_bwd_ws_bytes.argtypes = _lib._SIGS["snr_render_bwd_ws_bytes"][1]
ws_bytes = stamps.snr_render_bwd_ws_bytes(C.byref(a))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
d_rgb = torch.rand(N, 3, device=dev); d_depth = torch.rand(N, device=dev); d_acc = torch.rand(N, device=dev)
d_lat = torch.empty_like(lat); d_o = torch.zeros(N, 3, device=dev); d_d = torch.zeros(N, 3, device=dev)
dbg2 = torch.zeros(N * S, device=dev)
for _ in range(3):
    rc = stamps.snr_render_bwd(C.byref(a), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), d_rgb.data_ptr(), d_depth.data_ptr(), d_acc.data_ptr(),
                               d_lat.data_ptr(), d_o.data_ptr(), d_d.data_ptr(), dbg2.data_ptr(), ws.data_ptr(), ws_bytes,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
torch.cuda.synchronize()
tb = dbg2.cpu().numpy().view(np.uint64).reshape(-1, 16)[: N * S // 32].astype(np.int64)
bn = ["start", "composite bwd", "colour head", "rgb0^T"] + [f"layer {6 - i}^T done" for i in range(6)] + ["[enc_xyz^T: latent reduce done]", "(layers done)", "enc_xyz^T", "PE bwd", "end", "[enc_xyz^T: 8 of 16 steps]"]
db = tb - tb[:, :1]
print("backward, cycles from kernel start (median over %d wave tiles):" % len(tb))
prev = 0
for i, nme in sorted(enumerate(bn), key=lambda t: float(np.median(db[:, t[0]]))):      # in time order (stamps 10 and 15 sit inside enc_xyz^T)
    if not nme or not tb[:, i].any(): continue
    m = float(np.median(db[:, i]))
    print(f"  {nme:32s} {m:10.0f}  (+{m - prev:8.0f})")
    prev = m
