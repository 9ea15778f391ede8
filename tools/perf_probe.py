"""Quick timing probe of the fused forward kernel (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd
from supnerf_amd import ops
from oracle import supnerf_oracle as O

dev = torch.device("cuda:0")
params = O.init_decoder_params()
pk = ops.pack_weights({k: v.to(dev) for k, v in params.items()}, 3, 1)
for B, N, S in [(1, 1024, 64), (1, 4096, 64), (8, 4096, 64), (64, 4096, 64)]:
    g = torch.Generator().manual_seed(0)
    ro = (torch.randn(B * N, 3, generator=g) * 0.1 + torch.tensor([0., -12., 1.])).to(dev)
    vd = torch.randn(B * N, 3, generator=g) * 0.05 + torch.tensor([0., 1., 0.])
    vd = (vd / vd.norm(dim=-1, keepdim=True)).to(dev)
    z = (torch.linspace(9.5, 14.5, S)[None].repeat(B, 1)).to(dev)
    lat = torch.relu(torch.randn(B, 4, 256, generator=g)).to(dev) * 0.3
    div = torch.full((B,), 5.4, device=dev)
    cfg = ops.RenderCfg(S, ops.Z_PER_OBJECT, N, 3, 1)
    for _ in range(3):
        out = ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        out = ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    rays = B * N
    tf = rays * 57.56e6 / (ms * 1e-3) / 1e12
    print(f"B={B} N={N} S={S}: {ms:.3f} ms/launch  {rays/ms*1e3/1e6:.3f} Mrays/s  {tf:.1f} TFLOP/s ({tf/157.3*100:.1f}% of fp32 MFMA peak)", flush=True)
