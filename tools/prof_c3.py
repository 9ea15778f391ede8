"""Profiling target: BASELINE config 3's iteration -- 64 objects x 4096 rays x 64 samples in one launch per step (driver.optimize_objects_batched).
usage (under rocprofv3 --kernel-trace --stats): python3 tools/prof_c3.py [objects] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import driver as D, synthetic as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
hp = D.load_hpams(); hp["render_im_sz"] = 64; hp["optimize"]["num_opts"] = n_it
objs = D.make_objects(list(range(300, 300 + B)), 64)
g = torch.Generator().manual_seed(3)
sc, tc = torch.randn(B, 256, generator=g) * 0.3, torch.randn(B, 256, generator=g) * 0.3
D.optimize_objects_batched(model, dev, objs, hp, sc, tc, seeds=list(range(B))); torch.cuda.synchronize()
t0 = time.perf_counter()
D.optimize_objects_batched(model, dev, objs, hp, sc, tc, seeds=list(range(B))); torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"{B} objects per launch: {t / n_it * 1e3:.2f} ms/iteration incl. set-up ({B * n_it / t:.1f} object-iterations/s), {n_it} iterations")
