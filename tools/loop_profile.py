"""Where does the host time of the one-object optimise loop go?  (development aid)

Runs driver.optimize_object eagerly at 4096 x 64, prints ms/iteration, then a cProfile of the loop (host side only) and a
torch.profiler summary (number of device launches per iteration, top ops).  usage: python tools/loop_profile.py [iters]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import supnerf_amd as A
from supnerf_amd import driver as D, synthetic as O

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1)
model.load_state_dict(O.init_decoder_params())
model = model.to(dev)
hp = D.load_hpams()
hp["render_im_sz"] = 64
hp["optimize"]["num_opts"] = n_it
obj = D.make_objects([200], 64)[0]
g = torch.Generator().manual_seed(3)
sc, tc = torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3


def run():
    out = D.optimize_object(model, dev, obj, hp, sc, tc, seed=0)
    torch.cuda.synchronize()
    return out


run()
t0 = time.perf_counter(); run(); t = time.perf_counter() - t0
print(f"eager one-object loop: {t / n_it * 1e3:.3f} ms/iteration ({n_it / t:.1f} object-iterations/s)", flush=True)

pr = cProfile.Profile()
pr.enable(); run(); pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000], flush=True)

from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
ka = prof.key_averages()
n_kern = sum(e.count for e in ka if e.device_type.name == "CUDA" or getattr(e, "device_time_total", 0) > 0 and e.key.startswith(("void", "snr", "Memcpy", "Memset")))
print(ka.table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=70)[:12000])
print("launch-ish events per iteration:", n_kern / n_it)
