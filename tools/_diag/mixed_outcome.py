"""Experiment: which pieces of a training step need exact fp32 for a 60-step run to end where the reference's fp32 arithmetic ends?
(forward chain, backward chain, weight-gradient products) in every combination of interest, against the float64 oracle run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import supnerf_amd
from oracle import supnerf_oracle as O
T = supnerf_amd.trainer
dev = torch.device("cuda:0")
oracle_params = O.init_decoder_params(seed=0, sigma_bias=-2.0)
STEPS, B, n, S = int(os.environ.get("SNR_STEPS", "60")), int(os.environ.get("SNR_B", "2")), int(os.environ.get("SNR_N", "32")), 64
g = torch.Generator().manual_seed(5)
batches = [dict(code_idx=torch.tensor([(2 * k) % 6, (2 * k + 3) % 6]), xyz=torch.rand(B, n, S, 3, generator=g) - 0.5,
                viewdir=torch.nn.functional.normalize(torch.randn(B, n, 1, 3, generator=g), dim=-1).repeat(1, 1, S, 1),
                z_vals=torch.sort(torch.rand(B, S, generator=g) * 4 + 9, dim=-1)[0], rgb_tgt=torch.rand(B, n, 3, generator=g),
                occ_pixels=(torch.randint(0, 3, (B, n, 1), generator=g) - 1).float()) for k in range(4)]
hp = dict(lr_schedule=[dict(lr=1e-4, interval=40000), dict(lr=1e-4, interval=40000)])
def oracle_run(dtype, seed_codes=4):
    c = lambda t: t.to(dtype) if t.is_floating_point() else t
    p = {k: c(v).clone().requires_grad_() for k, v in oracle_params.items()}
    codes = T.CodeTables(6, 256, seed=seed_codes)
    w_sc, w_tc = c(codes.shape_codes.weight.detach()).clone().requires_grad_(), c(codes.texture_codes.weight.detach()).clone().requires_grad_()
    opt = torch.optim.AdamW([{"params": list(p.values()), "lr": 1e-4}, {"params": [w_sc], "lr": 1e-4}, {"params": [w_tc], "lr": 1e-4}])
    curve = []
    for it in range(STEPS):
        b = {k: c(v) for k, v in batches[it % 4].items()}
        opt.zero_grad()
        total = O.training_losses(p, b["xyz"], b["viewdir"], w_sc[b["code_idx"]], w_tc[b["code_idx"]], b["z_vals"], b["rgb_tgt"], b["occ_pixels"], 0.1)[0]
        total.backward(); opt.step(); curve.append(float(total))
        if it % 20 == 0: print(f"  oracle {dtype} step {it}", flush=True)
    return np.array(curve), {k: v.detach().double() for k, v in p.items()}
def gpu_run(precision):
    m = supnerf_amd.CodeNeRF(3, 1); m.load_state_dict(oracle_params, strict=True); m.precision = precision
    m = m.to(dev); m.train_decoder_weights = True
    codes = T.CodeTables(6, 256, seed=4).to(dev)
    bucket = T.GradBucket(list(m.parameters()) + list(codes.parameters()), row_sparse=list(codes.parameters()))
    opt = T.make_optimizer(m, codes, hp)
    dev_batches = [{k: v.to(dev) for k, v in b.items()} for b in batches]
    curve = [float(T.train_step(m, codes, opt, bucket, dev_batches[it % 4], 0.1)["loss_total"]) for it in range(STEPS)]
    return np.array(curve), {k: v.detach().double().cpu() for k, v in m.named_parameters()}
c64, w64 = oracle_run(torch.float64); c32, w32 = oracle_run(torch.float32)
init = {k: v.double() for k, v in oracle_params.items()}
wd = lambda w: max(float((w[k] - w64[k]).norm()) / (float((w64[k] - init[k]).norm()) + 1e-12) for k in w64)
print(f"fp32 oracle floor: loss curve {np.abs(c32 - c64).max():.2e}, weights {wd(w32):.2e}")
for precision in [tuple(a.split(",")) for a in sys.argv[1:]] or (("fp32", "fp32", "fp32"), ("fp32", "fp32", "bf16x3"), ("fp32", "bf16x3", "bf16x3"), ("fp32", "bf16x3", "fp32"), ("bf16x3", "fp32", "fp32"), ("bf16x3", "bf16x3", "bf16x3")):
    c, w = gpu_run(precision)
    print(f"forward chain {precision[0]:7s} backward chain {precision[1]:7s} products {precision[2]:7s}: loss curve vs float64 {np.abs(c - c64).max():.2e}, final weights {wd(w):.2e}", flush=True)
