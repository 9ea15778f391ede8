"""Experiment (VERDICT r3 #4b): may the training step's X / G dumps be NARROWER than fp32?  The split kernels' weight-gradient product
dW = G^T X reads 2 KB per point and layer (X and G as fp32) and is HBM-bound; X as one fp16 value and G as one bf16 value would halve
the dumps and the product's reads.  Emulated here WITHOUT touching a kernel: the fp32 dumps of the shipped step are rounded to the
narrower type on the device right before the products (monkey-patched ``ops.weight_grad``), then the 60-step outcome of
tests/test_driver_gpu.py::test_training_outcome_fp32_and_bf16x3_track_the_oracle is measured against the float64 oracle run.
usage: python tools/_diag/dump_width_outcome.py            (prints one line per combination)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import supnerf_amd
from supnerf_amd import ops
from oracle import supnerf_oracle as O
T = supnerf_amd.trainer
dev = torch.device("cuda:0")
oracle_params = O.init_decoder_params(seed=0, sigma_bias=-2.0)
STEPS, B, n, S = int(os.environ.get("SNR_STEPS", "60")), 2, 32, 64
g = torch.Generator().manual_seed(5)
batches = [dict(code_idx=torch.tensor([(2 * k) % 6, (2 * k + 3) % 6]), xyz=torch.rand(B, n, S, 3, generator=g) - 0.5,
                viewdir=torch.nn.functional.normalize(torch.randn(B, n, 1, 3, generator=g), dim=-1).repeat(1, 1, S, 1),
                z_vals=torch.sort(torch.rand(B, S, generator=g) * 4 + 9, dim=-1)[0], rgb_tgt=torch.rand(B, n, 3, generator=g),
                occ_pixels=(torch.randint(0, 3, (B, n, 1), generator=g) - 1).float()) for k in range(4)]
hp = dict(lr_schedule=[dict(lr=1e-4, interval=40000), dict(lr=1e-4, interval=40000)])

def oracle_run(dtype):
    c = lambda t: t.to(dtype) if t.is_floating_point() else t
    p = {k: c(v).clone().requires_grad_() for k, v in oracle_params.items()}
    codes = T.CodeTables(6, 256, seed=4)
    w_sc, w_tc = c(codes.shape_codes.weight.detach()).clone().requires_grad_(), c(codes.texture_codes.weight.detach()).clone().requires_grad_()
    opt = torch.optim.AdamW([{"params": list(p.values()), "lr": 1e-4}, {"params": [w_sc], "lr": 1e-4}, {"params": [w_tc], "lr": 1e-4}])
    curve = []
    for it in range(STEPS):
        b = {k: c(v) for k, v in batches[it % 4].items()}
        opt.zero_grad()
        total = O.training_losses(p, b["xyz"], b["viewdir"], w_sc[b["code_idx"]], w_tc[b["code_idx"]], b["z_vals"], b["rgb_tgt"], b["occ_pixels"], 0.1)[0]
        total.backward(); opt.step(); curve.append(float(total))
    return np.array(curve), {k: v.detach().double() for k, v in p.items()}

QX, QG = None, None
_real = ops.weight_grad
def q(t, kind):
    if kind is None: return t
    if kind == "f16": return t.half().float()
    if kind == "bf16": return t.bfloat16().float()
    if kind == "bf16x2":                      # hi + lo bf16 pieces (what the bf16x3 product keeps of an operand anyway)
        hi = t.bfloat16().float(); return hi + (t - hi).bfloat16().float()
    raise ValueError(kind)
def patched(G, n_out, X, n_in, want_bias=True, out=None, ws=None, precision="fp32"):
    if G.shape[1] >= 128:                     # (the chains' dumps; the two narrow heads' operands are not dumps)
        # db sums the UNROUNDED G in the kernels only if the dump keeps fp32; a narrower G dump feeds the bias sums too
        G = q(G.contiguous(), QG)
    X = q(X.contiguous(), QX) if X.shape[1] >= 64 else X
    return _real(G, n_out, X, n_in, want_bias=want_bias, out=out, ws=ws, precision=precision)
ops.weight_grad = patched

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
print(f"fp32 oracle floor ({STEPS} steps): loss curve {np.abs(c32 - c64).max():.2e}, weights {wd(w32):.2e}", flush=True)
for qx, qg in ((None, None), ("f16", None), (None, "bf16"), ("f16", "bf16"), ("bf16", "bf16"), ("bf16x2", "bf16x2"), ("f16", "bf16x2")):
    QX, QG = qx, qg
    c, w = gpu_run("auto")
    print(f"X dump {str(qx):7s} G dump {str(qg):7s}: loss curve vs float64 {np.abs(c - c64).max():.2e}, final weights {wd(w):.2e}", flush=True)
