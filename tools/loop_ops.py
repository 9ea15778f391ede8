"""Which torch ops stand behind the small launches of the fused one-object optimise loop (torch.profiler, CPU-side op names with their
device kernels).  usage: python tools/loop_ops.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import supnerf_amd as A
from supnerf_amd import driver as D, synthetic as O

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
hp = D.load_hpams(); hp["render_im_sz"] = 64; hp["optimize"]["num_opts"] = n_it
obj = D.make_objects([200], 64)[0]
g = torch.Generator().manual_seed(3)
sc, tc = torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3
D.optimize_object(model, dev, obj, hp, sc, tc, seed=0); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    D.optimize_object(model, dev, obj, hp, sc, tc, seed=0); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.cpu_time_total) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[1])
print(f"{n_it} iterations; op, calls, calls/iteration, device us total, cpu us total")
for k, c, d, cpu in rows[:60]:
    print(f"{k[:70]:70s} {c:6d} {c / n_it:7.2f} {d:10.0f} {cpu:10.0f}")
