"""Where the HOST time of the API-structured optimise loop goes (driver.optimize_object_api: the reference's loop body on the public
functions): cProfile over the loop, cumulative time per function, our package apart from torch.   usage: python tools/cprof_api_loop.py [iterations]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import driver as D, synthetic as O

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
hp = D.load_hpams(); hp["render_im_sz"] = 64; hp["optimize"]["num_opts"] = n_it
obj = D.make_objects([200], 64)[0]
g = torch.Generator().manual_seed(3)
sc, tc = torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3
D.optimize_object_api(model, dev, obj, hp, sc, tc, seed=0); torch.cuda.synchronize()
t0 = time.perf_counter()
D.optimize_object_api(model, dev, obj, hp, sc, tc, seed=0); torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"API-structured loop: {t / n_it * 1e3:.3f} ms/iteration, {n_it} iterations (no profiler)")
pr = cProfile.Profile(); pr.enable()
D.optimize_object_api(model, dev, obj, hp, sc, tc, seed=0); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("cumulative"); ps.print_stats(70)
print(s.getvalue())
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(45)
print(s.getvalue())
