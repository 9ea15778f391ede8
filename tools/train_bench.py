"""Training-step timing (BASELINE config 5 shape on one GPU): B objects x 1024 rays x 64 samples per step, decoder + codes trained."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import trainer as T, synthetic as SY

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PREC = sys.argv[2] if len(sys.argv) > 2 else "auto"      # auto | fp32 | bf16x3
n, S = 1024, 64
dev = torch.device("cuda:0")
m = A.CodeNeRF(3, 1); m.load_state_dict(SY.init_decoder_params()); m = m.to(dev); m.train_decoder_weights = True; m.precision = PREC
codes = T.CodeTables(64, 256, seed=1).to(dev)
g = torch.Generator().manual_seed(0)
batch = dict(code_idx=torch.arange(B), xyz=torch.rand(B, n, S, 3, generator=g) - 0.5,
             viewdir=torch.nn.functional.normalize(torch.randn(B, n, 1, 3, generator=g), dim=-1).repeat(1, 1, S, 1),
             z_vals=torch.sort(torch.rand(B, S, generator=g) * 4 + 9, dim=-1)[0], rgb_tgt=torch.rand(B, n, 3, generator=g),
             occ_pixels=(torch.randint(0, 3, (B, n, 1), generator=g) - 1).float())
batch = {k: v.to(dev) for k, v in batch.items()}
hp = dict(lr_schedule=[dict(lr=1e-4, interval=40000), dict(lr=1e-4, interval=40000)])
bucket = T.GradBucket(list(m.parameters()) + list(codes.parameters()), row_sparse=list(codes.parameters()))
opt = T.make_optimizer(m, codes, hp)
for _ in range(3):
    out = T.train_step(m, codes, opt, bucket, batch, 0.1)
torch.cuda.synchronize()
t0 = time.perf_counter()
k = 10
for _ in range(k):
    out = T.train_step(m, codes, opt, bucket, batch, 0.1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / k
print(f"train_step ({PREC}): B={B} objects x {n} rays x {S} samples: {dt*1e3:.2f} ms/step = {B*n/dt/1e3:.1f} k rays/s (loss {float(out['loss_total']):.4f}, "
      f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB)")
