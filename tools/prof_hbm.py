"""Profiling target: the HBM-bound stand-alone kernels at bench.py's shapes (run under rocprofv3 --kernel-trace --stats): encode with PE output,
composite forward (three input sets in turn: no help from the 256 MB Infinity Cache), scene composite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from supnerf_amd import ops, utils as U, synthetic as O

dev = torch.device("cuda:0")
N, S, B = 4096, 64, 64
ob = O.synthetic_object(100)
with torch.no_grad():
    ro, vd = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
    near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S))
ro_h, vd_h = ro.repeat(B, 1), vd.repeat(B, 1)
z_h = z[None].repeat(B, 1).contiguous()
div_h = torch.full((B,), float(ob["obj_diag"]), device=dev)
cfg = ops.RenderCfg(S, ops.Z_PER_OBJECT, N, 3, 1, frame=U._frame(False, False, True))
for _ in range(12):
    ops.encode(ro_h, vd_h, z_h, div_h, None, cfg, want_pe=True)
sets = [(torch.rand(B * N, S, device=dev), torch.rand(B * N, S, 3, device=dev)) for _ in range(3)]
for i in range(24):
    ops.composite_fwd(*sets[i % 3], z_h, ops.Z_PER_OBJECT, False, N)
del sets
P_s, n_s = 131072, 3 * S
strat = (torch.arange(S, device=dev) + torch.rand(P_s, 3, S, device=dev)) / S
z_s = (torch.rand(P_s, 3, 1, device=dev) * 8 + 2 + strat * 4).view(P_s, n_s)
sig_s, rgb_s = torch.rand(P_s, n_s, device=dev), torch.rand(P_s, n_s, 3, device=dev)
for _ in range(12):
    ops.scene_composite(sig_s, rgb_s, z_s, True, run_length=S)
torch.cuda.synchronize()
print(f"bytes per launch: encode {B * N * S * (12 + 12 + 4 + 63 * 4) + B * N * (24 + 27 * 4)}, composite_fwd {B * N * S * 16 + B * N * 20}, "
      f"scene {P_s * n_s * 20 + P_s * 20}")
