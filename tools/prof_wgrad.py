"""Profiling target: the weight-gradient product snr_weight_grad on a 256 x 256 layer at 524 288 points, operands rotating over 4 slots.
usage (under rocprofv3): python3 tools/prof_wgrad.py fp32|bf16x3 [launches]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import _lib
lib = _lib.lib()
prec = 0 if (sys.argv[1] if len(sys.argv) > 1 else "fp32") == "fp32" else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
P, NL = 8 * 1024 * 64, 4
G = torch.randn(NL, P, 256, device=dev).relu_(); X = torch.randn(NL, P, 256, device=dev).relu_()
dW = torch.empty(256, 256, device=dev); db = torch.empty(256, device=dev)
wsb = lib.snr_weight_grad_ws_bytes(P, 256, 256); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(n):
    l = i % NL
    assert lib.snr_weight_grad(G[l].data_ptr(), 256, 256, X[l].data_ptr(), 256, 256, P, dW.data_ptr(), 256, db.data_ptr(), prec, ws.data_ptr(), wsb, st) == 0
torch.cuda.synchronize()
print("ok", float(dW.abs().max()))
