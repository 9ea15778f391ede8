"""A/B timing of library variants (tools/build_variant.sh): forward, forward with ReLU bits and backward of the fused render at
4096 x 64, device-event times.  usage: python tools/ab_time.py NAME [NAME ...]  -- min and median over interleaved rounds   (NAME 'shipped' = the in-tree library)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import supnerf_amd as A
from supnerf_amd import _lib, ops, synthetic as SY, utils as U
good = _lib.lib()
dev = torch.device("cuda:0")
_sd = SY.init_decoder_params()
if os.environ.get("SNR_AB_SCALE"):        # power experiment: every decoder weight scaled (0 = all-zero operands: nothing toggles in the matrix pipe)
    _sd = {k: v * float(os.environ["SNR_AB_SCALE"]) for k, v in _sd.items()}
model = A.CodeNeRF(3, 1); model.load_state_dict(_sd); model = model.to(dev)
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
lb = None if os.environ.get("SNR_AB_NOLB") else model.latent_biases(lat)      # latent terms folded into the biases (as model.fused_render passes them)
a = ops._render_args(ro, vd, z, div, None, lat, pk, U._frame(False, False, True), 1.0, ops.Z_SHARED, 0, N, S, 3, 1, int(os.environ.get("SNR_AB_PRECISION", "1")),
                     latent_bias=lb)
rgb = torch.empty(N, 3, device=dev); depth = torch.empty(N, device=dev); acc = torch.empty(N, device=dev)
sig = torch.empty(N * S, device=dev); rgbs = torch.empty(N * S, 3, device=dev)
masks = torch.empty(int(good.snr_mask_bytes(N * S, 3, 1)), dtype=torch.uint8, device=dev)
d_rgb = torch.rand(N, 3, device=dev); d_depth = torch.rand(N, device=dev); d_acc = torch.rand(N, device=dev)
d_lat = torch.empty_like(lat); d_o = torch.zeros(N, 3, device=dev); d_d = torch.zeros(N, 3, device=dev)
a32 = ops._render_args(ro, vd, z, div, None, lat, pk, U._frame(False, False, True), 1.0, ops.Z_SHARED, 0, N, S, 3, 1, 0)
for fn in ("snr_render_fwd",):
    getattr(good, fn).restype, getattr(good, fn).argtypes = _lib._SIGS[fn]
rgb32 = torch.empty(N, 3, device=dev); depth32 = torch.empty(N, device=dev); acc32 = torch.empty(N, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
libs = {}
import copy
pp_names = ops.per_point_tensor_names(3, 1)
pp = [dict(model.named_parameters())[n].detach().contiguous() for n in pp_names]
a_shipped, keep = a, []
def setup(name):
    lib = good if name == "shipped" else C.CDLL(os.path.join(ROOT, "tools", "_diag", f"libvariant_{name}.so"), mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
    for fn in ("snr_render_fwd", "snr_render_bwd", "snr_render_bwd_ws_bytes", "snr_pack_weights", "snr_packed_bytes"):
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = _lib._SIGS[fn]
    # every variant packs the weights itself (the stream layouts may differ between variants)
    pk_v = torch.empty(lib.snr_packed_bytes(3, 1) // 4, device=dev)
    arr = (C.c_void_p * len(pp))(*[t.data_ptr() for t in pp])
    assert lib.snr_pack_weights(arr, len(pp), 3, 1, pk_v.data_ptr(), st()) == 0
    a_v = type(a_shipped)(); C.memmove(C.byref(a_v), C.byref(a_shipped), C.sizeof(a_shipped)); a_v.packed = pk_v.data_ptr(); keep.append(pk_v)
    wsb = lib.snr_render_bwd_ws_bytes(C.byref(a_v)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    f0 = lambda: lib.snr_render_fwd(C.byref(a_v), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), None, None, None, st())
    f1 = lambda: lib.snr_render_fwd(C.byref(a_v), rgb.data_ptr(), depth.data_ptr(), acc.data_ptr(), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), st())
    b = lambda: lib.snr_render_bwd(C.byref(a_v), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), d_rgb.data_ptr(), d_depth.data_ptr(), d_acc.data_ptr(),
                                   (None if os.environ.get('SNR_AB_NOLAT') else d_lat.data_ptr()), d_o.data_ptr(), d_d.data_ptr(), None, ws.data_ptr(), wsb, st())
    assert f1() == 0 and b() == 0
    good.snr_render_fwd(C.byref(a32), rgb32.data_ptr(), depth32.data_ptr(), acc32.data_ptr(), None, None, None, st())
    f0(); torch.cuda.synchronize()
    err = f"rgb {float((rgb - rgb32).abs().max()):.1e} depth {float((depth - depth32).abs().max()):.1e} vs fp32"
    if name != "shipped" and int(os.environ.get("SNR_AB_PRECISION", "1")) == 0:
        # the variant's saved ReLU bits against the shipped kernel's (same layout; bits of units within rounding of zero may differ)
        m_v = masks.clone(); s_v = sig.clone()
        a_g = type(a_shipped)(); C.memmove(C.byref(a_g), C.byref(a_shipped), C.sizeof(a_shipped))
        good.snr_render_fwd(C.byref(a_g), rgb32.data_ptr(), depth32.data_ptr(), acc32.data_ptr(), sig.data_ptr(), rgbs.data_ptr(), masks.data_ptr(), st())
        torch.cuda.synchronize()
        x = (m_v ^ masks); nbits = int(sum(((x >> k) & 1).sum() for k in range(8)))
        err += f"; ReLU bits differing from the shipped kernel's: {nbits} of {masks.numel() * 8} ({nbits / (masks.numel() * 8):.2e}); sigma {float((s_v - sig).abs().max()):.1e}"
        f1(); torch.cuda.synchronize()
    b(); torch.cuda.synchronize()
    return f0, f1, b, err, ws, (d_lat.clone(), d_o.clone(), d_d.clone())
names = list(dict.fromkeys(sys.argv[1:]))
for n in names: libs[n] = setup(n)
res = {n: [[], [], []] for n in names}
for rnd in range(int(os.environ.get("SNR_AB_ROUNDS", "5"))):          # interleaved rounds: clock drift hits every variant alike
    for n in names:
        for k in range(3): res[n][k].append(timed(libs[n][k], 20))
for n in names[1:]:          # backward outputs against the first name's
    ref, got = libs[names[0]][5], libs[n][5]
    print(n, "backward vs", names[0], " ".join(f"{k} {float((g_ - r_).abs().max() / r_.abs().max()):.1e}" for k, r_, g_ in zip(("d_latent", "d_rays_o", "d_rays_d"), ref, got)), flush=True)
for n in names:
    r = res[n]
    print(f"{n:14s} fwd {min(r[0]):.4f} (med {sorted(r[0])[len(r[0]) // 2]:.4f})   fwd+bits {min(r[1]):.4f} (med {sorted(r[1])[len(r[1]) // 2]:.4f})   "
          f"bwd {min(r[2]):.4f} (med {sorted(r[2])[len(r[2]) // 2]:.4f}) ms   [{libs[n][3]}]", flush=True)
