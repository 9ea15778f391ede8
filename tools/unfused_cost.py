"""What the three-launch path (encode -> decoder -> composite, taken when n_samples does not divide 128) costs per sample point against the
fused single launch: the same 4096 rays at S = 64 and 128 (fused) and S = 96, 48, 65 (three launches).  usage: python tools/unfused_cost.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import supnerf_amd as A
from supnerf_amd import ops, synthetic as SY, utils as U
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev)
ob = SY.synthetic_object(100)
g = torch.Generator().manual_seed(100)
sc = (torch.randn(1, 256, generator=g) * 0.3).to(dev); tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for prec in ("fp32", "auto"):
    model.precision = prec
    for S in (64, 128, 96, 48, 65):
        with torch.no_grad():
            ro, vd = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
            near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
            z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S, generator=g))
            call = lambda: U._render_shared_z(model, dev, ro, vd, z, float(ob["obj_diag"]), U._frame(False, False, True), sc, tc)
            call()
            for _ in range(3): call()
            times = []
            for _ in range(10):
                torch.cuda.synchronize(); e0.record(); call(); e1.record(); torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            ms = sorted(times)[len(times) // 2]
            if max(times) > 2 * ms: print("   (outliers:", " ".join(f"{t:.2f}" for t in times), ")")
        print(f"{prec:5s} S={S:4d}: {ms:.3f} ms  {ms * 1e6 / (4096 * S):.2f} ns/point  ({'fused' if 128 % S == 0 else 'three launches'})")
