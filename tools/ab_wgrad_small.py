"""A/B timing of library variants on the two narrow heads' weight-gradient products (density head 1 x 256, colour head 3 x 128 of a 256-wide
dump) at 524 288 points.   usage: python tools/ab_wgrad_small.py NAME [NAME ...]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import supnerf_amd as A
from supnerf_amd import _lib
good = _lib.lib()
dev = torch.device("cuda:0")
P, NL = 8 * 1024 * 64, 4
X = torch.randn(NL, P, 256, device=dev); G1 = torch.randn(NL, P, 1, device=dev); G3 = torch.randn(NL, P, 3, device=dev)
dW = torch.empty(3, 256, device=dev); db = torch.empty(3, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
keep = []
def setup(name):
    lib = good if name == "shipped" else C.CDLL(os.path.join(ROOT, "tools", "_diag", f"libvariant_{name}.so"), mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
    for fn in ("snr_weight_grad", "snr_weight_grad_ws_bytes"):
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = _lib._SIGS[fn]
    wsb = lib.snr_weight_grad_ws_bytes(P, 3, 256); ws = torch.empty(wsb, dtype=torch.uint8, device=dev); keep.append(ws)
    def run():
        for l in range(NL):
            assert lib.snr_weight_grad(G1[l].data_ptr(), 1, 1, X[l].data_ptr(), 256, 256, P, dW.data_ptr(), 256, db.data_ptr(), 0, ws.data_ptr(), wsb, st()) == 0
            assert lib.snr_weight_grad(G3[l].data_ptr(), 3, 3, X[l].data_ptr(), 256, 128, P, dW.data_ptr(), 256, db.data_ptr(), 0, ws.data_ptr(), wsb, st()) == 0
    run(); torch.cuda.synchronize()
    ref = (G3[NL - 1].double().t() @ X[NL - 1][:, :128].double())
    return run, float((dW[:, :128].double() - ref).abs().max() / ref.abs().max())
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / NL
names = list(dict.fromkeys(sys.argv[1:]))
libs = {n: setup(n) for n in names}
res = {n: [] for n in names}
for rnd in range(4):
    for n in names: res[n].append(timed(libs[n][0]))
for n in names:
    print(f"{n:12s} both heads {min(res[n]) * 1e3:.1f} us (med {sorted(res[n])[2] * 1e3:.1f})   [colour head vs float64: {libs[n][1]:.1e}]", flush=True)
