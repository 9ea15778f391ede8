"""Development probe: forward time vs number of 256-wide layers (per-layer cost and fixed overhead)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, synthetic as SY, utils as U
dev = torch.device("cuda:0")
N, S = 4096, 64
ob = SY.synthetic_object(100)
g = torch.Generator().manual_seed(100)
with torch.no_grad():
    ro, vd = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[64, 64])
    near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
    z = U._shared_depths(near, far, S, dev, jitter=torch.rand(S, generator=g))
div = torch.full((1,), float(ob["obj_diag"]), device=dev)
for prec in sys.argv[1:] or ["bf16x3"]:
    for sb, tb in [(0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (2, 2)]:
        model = A.CodeNeRF(sb, tb); model.load_state_dict(SY.init_decoder_params(sb, tb)); model = model.to(dev)
        lat = torch.rand(1, max(sb + tb, 1), 256, device=dev)
        pk = model.packed_weights()
        cfg = ops.RenderCfg(S, ops.Z_SHARED, N, sb, tb, precision=prec)
        for _ in range(3): ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.render_fwd(ro, vd, z, div, None, lat, pk, cfg)
        e1.record(); torch.cuda.synchronize()
        print(f"{prec} sb={sb} tb={tb}: {e0.elapsed_time(e1)/30*1000:.1f} us", flush=True)
