"""Would the training step gain from keeping the activation / gradient dumps in the 256 MB Infinity Cache?  (VERDICT r2 #7b)

The training step writes every MFMA layer's input X_l (forward chain) and pre-activation gradient G_l (backward chain) to HBM at
1 KB / point / layer and reads them back in the weight-gradient products: 8 layers x 524 288 points x 1 KB x 2 (X, G) x 2 (write, read)
= 17 GB per step.  Processing the batch in chunks small enough for X + G of a chunk to stay cache-resident (chain forward, chain backward
and the nine products per chunk, dW accumulated over the chunks) would remove most of that traffic -- but a chunk of c points is only
c / 128 workgroups for the chain kernels, and the chip needs >= 256 of them.  This probe measures it: the decoder part of the step
(forward chain with dumps, backward chain with dumps, weight-gradient products) on 524 288 points in ONE pass and in chunks of
8 K ... 256 K points, device events, both arithmetics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, synthetic as O

dev = torch.device("cuda:0")
P, B = 8 * 1024 * 64, 8
params = O.init_decoder_params(seed=0)
model = A.CodeNeRF(3, 1); model.load_state_dict(params); model = model.to(dev)
g = torch.Generator().manual_seed(0)
xyz = (torch.rand(P, 3, generator=g) - 0.5).to(dev)
vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
lat = torch.rand(B, 4, 256, generator=g).to(dev)
d_sig, d_rgb = torch.rand(P, generator=g).to(dev), torch.rand(P, 3, generator=g).to(dev)
names = ops.per_point_tensor_names(3, 1)
weights = [dict(model.named_parameters())[n].detach() for n in names]


def step(chunk, prec):
    """decoder forward + backward + weight gradients over all P points, `chunk` points at a time (whole objects per chunk when chunk >= P / B,
    else one object's points split: every chunk belongs to one object)."""
    acc = None
    ppo = P // B
    for s in range(0, P, chunk):
        e = s + chunk
        lat_c = lat[s // ppo:(e - 1) // ppo + 1]
        x, v = xyz[s:e].clone().requires_grad_(), vd[s:e]
        w = [t.clone().requires_grad_() for t in weights]
        sig, rgb = ops.DecoderPointsTrain.apply(x, v, lat_c.clone().requires_grad_(), 3, 1, prec, *w)
        torch.autograd.backward([sig, rgb], [d_sig[s:e], d_rgb[s:e]])
        grads = [t.grad for t in w]
        acc = grads if acc is None else [a + b for a, b in zip(acc, grads)]
    return acc


def ev(fn, n=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for prec in ("bf16x3", "fp32"):
    base = None
    for chunk in (P, P // 2, P // 4, P // 8, P // 16, P // 32, P // 64):
        ms = ev(lambda: step(chunk, prec))
        base = base or ms
        dump_mb = 2 * 8 * chunk * 1024 / 2**20
        print(f"{prec}: chunks of {chunk:7d} points ({chunk // 128:5d} workgroups, X+G dumps {dump_mb:7.0f} MB): {ms:7.3f} ms  ({ms / base:.2f}x the one-pass step)", flush=True)
