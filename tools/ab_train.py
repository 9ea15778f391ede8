"""A/B timing of library variants (tools/build_variant.sh) on the TRAINING chains: snr_decoder_fwd with ReLU bits (+ activation dumps) and
snr_decoder_bwd (+ gradient dumps) at config 5's per-GPU shape (8 objects x 1024 rays x 64 samples = 524 288 points), device-event times.
usage: python tools/ab_train.py NAME [NAME ...]   (NAME 'shipped' = the in-tree library; SNR_AB_PRECISION=0 fp32 (default) / 1 split-bf16)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import supnerf_amd as A
from supnerf_amd import _lib, ops, synthetic as SY
good = _lib.lib()
dev = torch.device("cuda:0")
prec = int(os.environ.get("SNR_AB_PRECISION", "0"))
P, B, SB, TB = 8 * 1024 * 64, 8, 3, 1
model = A.CodeNeRF(SB, TB); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev)
g = torch.Generator().manual_seed(0)
xyz = (torch.rand(P, 3, generator=g) - 0.5).to(dev)
vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
lat = torch.rand(B, 4, 256, generator=g).to(dev)
d_sig, d_rgb = torch.rand(P, generator=g).to(dev), torch.rand(P, 3, generator=g).to(dev)
NL = SB + TB + 4
sig = torch.empty(P, device=dev); rgbs = torch.empty(P, 3, device=dev)
masks = torch.empty(int(good.snr_mask_bytes(P, SB, TB)), dtype=torch.uint8, device=dev)
act = torch.empty(NL, P, 256, device=dev); gd = torch.empty(NL, P, 256, device=dev)
d_lat = torch.empty_like(lat); d_x = torch.empty(P, 3, device=dev); d_v = torch.empty(P, 3, device=dev)
pp = [dict(model.named_parameters())[n].detach().contiguous() for n in ops.per_point_tensor_names(SB, TB)]
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
keep = []
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def setup(name):
    lib = good if name == "shipped" else C.CDLL(os.path.join(ROOT, "tools", "_diag", f"libvariant_{name}.so"), mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
    for fn in ("snr_decoder_fwd", "snr_decoder_bwd", "snr_decoder_bwd_ws_bytes", "snr_pack_weights", "snr_packed_bytes"):
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = _lib._SIGS[fn]
    pk = torch.empty(lib.snr_packed_bytes(SB, TB) // 4, device=dev)
    arr = (C.c_void_p * len(pp))(*[t.data_ptr() for t in pp])
    assert lib.snr_pack_weights(arr, len(pp), SB, TB, pk.data_ptr(), st()) == 0
    wsb = lib.snr_decoder_bwd_ws_bytes(P, P // B, SB, TB); ws = torch.empty(wsb, dtype=torch.uint8, device=dev); keep.extend([pk, ws])
    f = lambda a: lib.snr_decoder_fwd(xyz.data_ptr(), vd.data_ptr(), lat.data_ptr(), pk.data_ptr(), P, P // B, SB, TB, sig.data_ptr(), rgbs.data_ptr(),
                                      masks.data_ptr(), a, prec, st())
    b = lambda gdp: lib.snr_decoder_bwd(xyz.data_ptr(), vd.data_ptr(), lat.data_ptr(), pk.data_ptr(), masks.data_ptr(), sig.data_ptr(), d_sig.data_ptr(), d_rgb.data_ptr(),
                                        P, P // B, SB, TB, d_lat.data_ptr(), d_x.data_ptr(), d_v.data_ptr(), gdp, ws.data_ptr(), wsb, prec, st())
    assert f(act.data_ptr()) == 0 and b(gd.data_ptr()) == 0
    return (lambda: f(None)), (lambda: f(act.data_ptr())), (lambda: b(None)), (lambda: b(gd.data_ptr()))
names = list(dict.fromkeys(sys.argv[1:]))
libs = {n: setup(n) for n in names}
res = {n: [[], [], [], []] for n in names}
for rnd in range(int(os.environ.get("SNR_AB_ROUNDS", "4"))):
    for n in names:
        for k in range(4): res[n][k].append(timed(libs[n][k]))
for n in names:
    r = res[n]
    print(f"{n:14s} fwd+bits {min(r[0]):.3f}   fwd+bits+dump {min(r[1]):.3f}   bwd {min(r[2]):.3f}   bwd+dump {min(r[3]):.3f} ms   (precision {prec}, {P} points)", flush=True)
