"""Profiling target: the one-object fused optimise loop at 4096 x 64 (run under rocprofv3 --kernel-trace --stats): how many launches an
iteration is and what each costs.  usage: python tools/prof_loop.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import driver as D, synthetic as O

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
hp = D.load_hpams(); hp["render_im_sz"] = 64; hp["optimize"]["num_opts"] = n_it
obj = D.make_objects([200], 64)[0]
g = torch.Generator().manual_seed(3)
sc, tc = torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3
D.optimize_object(model, dev, obj, hp, sc, tc, seed=0); torch.cuda.synchronize()
t0 = time.perf_counter()
D.optimize_object(model, dev, obj, hp, sc, tc, seed=0); torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"fused one-object loop: {t / n_it * 1e3:.3f} ms/iteration ({n_it / t:.1f} object-iterations/s), {n_it} iterations")
