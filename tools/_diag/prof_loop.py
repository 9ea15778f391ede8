"""Host-side timing of the optimise loops (development aid): one-object loop, batched loop, batched loop replayed as a HIP graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import supnerf_amd as A
from supnerf_amd import driver as D, synthetic as SY
dev = torch.device("cuda:0")
model = A.CodeNeRF(3, 1); model.load_state_dict(SY.init_decoder_params()); model = model.to(dev)
hp = D.load_hpams(); hp["render_im_sz"] = 64; hp["optimize"]["num_opts"] = 20
for B in (1, 8):
    objs = D.make_objects(list(range(200, 200 + B)), 64)
    g = torch.Generator().manual_seed(3)
    sc, tc = torch.randn(B, 256, generator=g) * 0.3, torch.randn(B, 256, generator=g) * 0.3
    res = {}
    for name, fn in (("batched eager", lambda: D.optimize_objects_batched(model, dev, objs, hp, sc, tc, list(range(B)))),
                     ("batched graph", lambda: D.optimize_objects_batched(model, dev, objs, hp, sc, tc, list(range(B)), graph=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        res[name] = out
        print(f"B={B} {name}: 20 its {round((time.perf_counter() - t0) * 1e3, 1)} ms")
    m0, m1 = res["batched eager"][0], res["batched graph"][0]
    print("   metrics max |diff|:", [float((m0[..., k] - m1[..., k]).abs().max()) for k in range(4)],
          "codes:", float((res["batched eager"][1] - res["batched graph"][1]).abs().max()))
