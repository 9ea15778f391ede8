"""A/B timing of the scene composite kernel across library variants (tools/build_variant.sh NAME -D...): 131072 pixels x 3 lists x 64
stratified depths.  usage: python tools/scene_time.py NAME [NAME ...]   ('shipped' = the in-tree library)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from supnerf_amd import _lib
good = _lib.lib()
dev = torch.device("cuda:0")
P, NB, S = 131072, 3, 64
n = NB * S
g = torch.Generator(device=dev).manual_seed(1)
strat = (torch.arange(S, device=dev) + torch.rand(P, NB, S, device=dev, generator=g)) / S
z = (torch.rand(P, NB, 1, device=dev, generator=g) * 8 + 2 + strat * 4).view(P, n).contiguous()
sig = torch.rand(P, n, device=dev, generator=g); rgb = torch.rand(P, n, 3, device=dev, generator=g)
o_rgb = torch.empty(P, 3, device=dev); o_d = torch.empty(P, device=dev); o_a = torch.empty(P, device=dev)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ref = None
for name in sys.argv[1:]:
    lib = good if name == "shipped" else C.CDLL(os.path.join(ROOT, "tools", "_diag", f"libvariant_{name}.so"), mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
    lib.snr_scene_composite_fwd.restype, lib.snr_scene_composite_fwd.argtypes = _lib._SIGS["snr_scene_composite_fwd"]
    fn = lambda: lib.snr_scene_composite_fwd(sig.data_ptr(), rgb.data_ptr(), z.data_ptr(), P, n, S, 1, o_rgb.data_ptr(), o_d.data_ptr(), o_a.data_ptr(), st())
    assert fn() == 0
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    if ref is None: ref = o_rgb.clone()
    print(f"{name:16s} {ms:.4f} ms  {(P * n * 20 + P * 20) / ms / 1e6:.0f} GB/s   max |rgb - first| {float((o_rgb - ref).abs().max()):.1e}")
