"""Profiling target: the public render functions at 4096 x 64 (run under rocprofv3 --kernel-trace --stats): which launches a call of
utils.render_rays_v2 / NeRFRenderer.render_rays (forward, and forward + loss tail + backward to codes and pose) consists of.
usage: python tools/prof_api.py [a|b] [fwd|bwd] [calls] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, utils as U, synthetic as O

fam = sys.argv[1] if len(sys.argv) > 1 else "a"
mode = sys.argv[2] if len(sys.argv) > 2 else "bwd"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
prec = sys.argv[4] if len(sys.argv) > 4 else "auto"
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev); model.precision = prec
ob = O.synthetic_object(100)
img, mask = O.synthetic_targets(100, 64)
g = torch.Generator().manual_seed(100)
sc = (torch.randn(1, 256, generator=g) * 0.3).to(dev).requires_grad_(); tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev).requires_grad_()
pose = ob["cam_pose"].to(dev).requires_grad_()
rend = A.NeRFRenderer(n_samples=64, white_bkgd=True)


def call():
    if fam == "a":
        return U.render_rays_v2(model, dev, img, mask, pose, ob["obj_diag"], ob["K"], ob["roi"], 64, sc, tc, 1, 0, im_sz=64)
    return rend.render_rays(model, dev, img, mask, pose, ob["wlh"], ob["K"], ob["roi"], sc, tc, im_sz=64)


def step():
    if mode == "fwd":
        with torch.no_grad():
            call()
    else:
        out = call()
        loss, _ = ops.LossTail.apply(out[0], out[2], out[3], out[4], 0.1, 4096)
        sc.grad = tc.grad = pose.grad = None
        loss.sum().backward()


for _ in range(5):
    step()
batches = []
for b in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    batches.append((time.perf_counter() - t0) / n * 1e3)
print(f"family {fam} {mode} {prec}: {min(batches):.3f} ms per call (best of 5 batches of {n} calls: {' '.join(f'{t:.3f}' for t in batches)})")
