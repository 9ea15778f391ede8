"""A/B timing of library variants (tools/build_variant.sh) on the weight-gradient product snr_weight_grad: 256 x 256 layer, 524 288 points
(config 5's per-GPU shape), operands rotating over 8 layer slots like the training step's dumps; device-event time per product.
usage: python tools/ab_wgrad.py NAME [NAME ...]   (NAME 'shipped' = the in-tree library)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import supnerf_amd as A
from supnerf_amd import _lib
good = _lib.lib()
dev = torch.device("cuda:0")
P, NL = 8 * 1024 * 64, 8
G = torch.randn(NL, P, 256, device=dev); X = torch.randn(NL, P, 256, device=dev)
dW = torch.empty(256, 256, device=dev); db = torch.empty(256, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
keep = []
def setup(name):
    lib = good if name == "shipped" else C.CDLL(os.path.join(ROOT, "tools", "_diag", f"libvariant_{name}.so"), mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
    for fn in ("snr_weight_grad", "snr_weight_grad_ws_bytes"):
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = _lib._SIGS[fn]
    wsb = lib.snr_weight_grad_ws_bytes(P, 256, 256); ws = torch.empty(wsb, dtype=torch.uint8, device=dev); keep.append(ws)
    def run(prec):
        for l in range(NL):
            rc = lib.snr_weight_grad(G[l].data_ptr(), 256, 256, X[l].data_ptr(), 256, 256, P, dW.data_ptr(), 256, db.data_ptr(), prec, ws.data_ptr(), wsb, st())
            assert rc == 0
    ref = (G[NL - 1].double().t() @ X[NL - 1].double())
    run(0); torch.cuda.synchronize()
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    return run, err
def timed(fn, prec, n=3):
    fn(prec); torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn(prec)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / NL
names = list(dict.fromkeys(sys.argv[1:]))
libs = {n: setup(n) for n in names}
res = {n: [[], []] for n in names}
for rnd in range(int(os.environ.get("SNR_AB_ROUNDS", "4"))):
    for n in names:
        for prec in (0, 1): res[n][prec].append(timed(libs[n][0], prec))
for n in names:
    r = res[n]
    print(f"{n:14s} fp32 {min(r[0]):.4f} (med {sorted(r[0])[len(r[0]) // 2]:.4f})   bf16x3 {min(r[1]):.4f} (med {sorted(r[1])[len(r[1]) // 2]:.4f}) ms per 256x256 product at {P} points   "
          f"[fp32 product vs float64: {libs[n][1]:.1e}]", flush=True)
