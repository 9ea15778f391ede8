"""How many calls of the public render functions does a fresh process need to reach its steady state?  Per-call wall times (synchronised)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, utils as U, synthetic as O
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
ob = O.synthetic_object(100); img, mask = O.synthetic_targets(100, 64)
g = torch.Generator().manual_seed(100)
sc = (torch.randn(1, 256, generator=g) * 0.3).to(dev); tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev)
pose = ob["cam_pose"].to(dev)
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        U.render_rays_v2(model, dev, img, mask, pose, ob["obj_diag"], ob["K"], ob["roi"], 64, sc, tc, 1, 0, im_sz=64)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("synchronised per-call ms:", " ".join(f"{t:.2f}" for t in ts))
# unsynchronised batches of 10
for b in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        with torch.no_grad():
            U.render_rays_v2(model, dev, img, mask, pose, ob["obj_diag"], ob["K"], ob["roi"], 64, sc, tc, 1, 0, im_sz=64)
    torch.cuda.synchronize(); print(f"batch {b}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call")
