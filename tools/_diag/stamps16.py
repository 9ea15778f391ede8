"""Per-phase timeline of the two-waves-per-SIMD fp32 kernels (snr_mlp16.hip, snr_mlp16_bwd.hip) from in-kernel s_memtime stamps.
Build: tools/build_diag.sh s16   ->  tools/_diag/libsupnerf_stamps_s16.so ; run: python tools/_diag/stamps16.py"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import supnerf_amd as A
from supnerf_amd import _lib, ops, synthetic as SY, utils as U
good = _lib.lib()
st_lib = C.CDLL(os.path.join(ROOT, "tools", "_diag", os.environ.get("SNR_STAMP_LIB", "libsupnerf_stamps_s16.so")), mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
for fn in ("snr_render_fwd", "snr_render_bwd", "snr_render_bwd_ws_bytes"):
    getattr(st_lib, fn).restype, getattr(st_lib, fn).argtypes = _lib._SIGS[fn]
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev)
N, S = int(os.environ.get("SNR_STAMP_RAYS", "4096")), 64
ob = SY.synthetic_object(100)
g = torch.Generator().manual_seed(100)
sc = (torch.randn(1, 256, generator=g) * 0.3).to(dev); tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev)
with torch.no_grad():
    ro, vd = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
    near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S, generator=g))
    lat = model.latent_terms(sc, tc).contiguous()
ro, vd = ro[:N].contiguous(), vd[:N].contiguous()
pk = model.packed_weights()
div = torch.full((1,), float(ob["obj_diag"]), device=dev)
a = ops._render_args(ro, vd, z, div, None, lat, pk, U._frame(False, False, True), 1.0, ops.Z_SHARED, 0, N, S, 3, 1, 0, latent_bias=model.latent_biases(lat))
rgb = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev); acc = torch.empty(N, device=dev)
stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def slots(raw):
    """forward stamps carry HW_ID | XCC_ID << 32 in slot 6: pair every wave with the wave that ran before it on the same SIMD"""
    hw = raw[:, 6].astype(np.uint64)
    lo = (hw & np.uint64(0xffffffff)).astype(np.int64); xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    simd = (lo >> 4) & 3; cu = (lo >> 8) & 15; sh = (lo >> 12) & 1; se = (lo >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    t = raw.astype(np.int64)
    gaps, alone = [], []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        st, en = t[idx, 0], t[idx, 7]
        order = np.argsort(st)
        st, en = st[order], en[order]
        for j in range(2, len(st)):                 # the first two waves of a SIMD start together
            prev_end = en[:j][en[:j] <= st[j]]
            if len(prev_end): gaps.append(st[j] - prev_end.max())
    if not gaps: return
    print(f"  SIMDs seen: {len(np.unique(key))}, waves per SIMD: {len(key) / len(np.unique(key)):.1f}; start of a wave minus the end of the wave it replaces on its SIMD: "
          f"median {np.median(gaps):.0f}, p10 {np.percentile(gaps, 10):.0f}, p90 {np.percentile(gaps, 90):.0f} cycles ({len(gaps)} pairs)")

def report(title, t, names):
    t = t.astype(np.int64)
    d = t - t[:, :1]
    print(f"== {title}: cycles from the wave's start, median over {len(t)} waves (p10 / p90)")
    prev = 0.0
    for i, n in enumerate(names):
        m = float(np.median(d[:, i]))
        print(f"  {n:28s} {m:9.0f}  (+{m - prev:8.0f})   {np.percentile(d[:, i], 10):9.0f} / {np.percentile(d[:, i], 90):9.0f}")
        prev = m
    t0 = t[:, 0] - t[:, 0].min(); t1 = t[:, -1] - t[:, 0].min()
    life = float(np.median(t1 - t0))
    print(f"  wave lifetime median {life:.0f}, p10 {np.percentile(t1 - t0, 10):.0f}, p90 {np.percentile(t1 - t0, 90):.0f}")

dbg = torch.zeros(N * S, device=dev)
for _ in range(3):
    assert st_lib.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), dbg.data_ptr(), None, None, stream()) == 0
torch.cuda.synchronize()
fw_names = ["start", "prologue done", "enc_xyz done", "shape layers + enc_shape", "enc_viewdir + texture", "rgb.0 done", "colour head done", "end (composite)"]
raw = dbg.cpu().numpy().view(np.uint64).reshape(-1, 8)[: N * S // 16]
slots(raw)
raw = raw.copy(); raw[:, 6] = raw[:, 5]
report("forward, no ReLU bits saved", raw, fw_names)
masks_dbg = torch.empty(int(good.snr_mask_bytes(N * S, 3, 1)), dtype=torch.uint8, device=dev); rgbs_dbg = torch.empty(N * S, 3, device=dev)
dbg.zero_()
for _ in range(3):
    assert st_lib.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), dbg.data_ptr(), rgbs_dbg.data_ptr(), masks_dbg.data_ptr(), stream()) == 0
torch.cuda.synchronize()
raw = dbg.cpu().numpy().view(np.uint64).reshape(-1, 8)[: N * S // 16]
slots(raw)
raw = raw.copy(); raw[:, 6] = raw[:, 5]
report("forward saving the ReLU bits", raw, fw_names)

sig = torch.empty(N * S, device=dev); rgbs = torch.empty(N * S, 3, device=dev)
masks = torch.empty(int(good.snr_mask_bytes(N * S, 3, 1)), dtype=torch.uint8, device=dev)
assert good.snr_render_fwd(C.byref(a), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), stream()) == 0
wsb = st_lib.snr_render_bwd_ws_bytes(C.byref(a)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
d_rgb = torch.rand(N, 3, device=dev); d_depth = torch.rand(N, device=dev); d_acc = torch.rand(N, device=dev)
d_lat = torch.empty_like(lat); d_o = torch.zeros(N, 3, device=dev); d_d = torch.zeros(N, 3, device=dev)
dbg2 = torch.zeros(N * S, device=dev)
for _ in range(3):
    assert st_lib.snr_render_bwd(C.byref(a), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), d_rgb.data_ptr(), d_depth.data_ptr(), d_acc.data_ptr(),
                                 d_lat.data_ptr(), d_o.data_ptr(), d_d.data_ptr(), dbg2.data_ptr(), ws.data_ptr(), wsb, stream()) == 0
torch.cuda.synchronize()
rawb = dbg2.cpu().numpy().view(np.uint64).reshape(-1, 8)[: N * S // 16].astype(np.int64)
if os.environ.get("SNR_STAMP_TAIL"):      # library built with -DSNR16_TAILSTAMPS: slots 3, 4 lie inside the tail
    for a_, b_, nm in ((6, 3, "enc_xyz^T done -> scratch written + barrier"), (3, 4, "-> encoding gradient done"), (4, 7, "-> end (ray tail)")):
        dd = rawb[:, b_] - rawb[:, a_]
        print(f"  tail: {nm:46s} median {np.median(dd):8.0f}  p10 {np.percentile(dd, 10):8.0f}  p90 {np.percentile(dd, 90):8.0f}")
    rawb[:, 3] = rawb[:, 2]; rawb[:, 4] = rawb[:, 2]
report("backward", rawb,
       ["start", "composite backward done", "colour head done", "rgb.0^T done", "texture^T + enc_viewdir^T", "enc_shape^T + shape^T", "enc_xyz^T done", "end (encoding, ray tail)"])
print("per wave: 7040 MFMAs x 32 cycles = 225280 cycles of matrix work; two waves share a SIMD's pipe")
