"""Operand-truncation error of split-operand matrix products, emulated on the CPU (exact products of the pieces, float64 accumulation, so
that only the splitting shows): the shipped bf16 two-piece split (3 products), an fp16 two-piece split (3 products, optionally with
power-of-two scales on the operands so that the low pieces stay normal), a bf16 three-piece split (6 products), against the float64
product and torch's fp32 matmul.  A (4096 x 256) activation matrix with ReLU zeros and a wide per-row dynamic range times a 256 x 256
weight matrix at the decoder's init scale.   usage: python tools/_diag/split_accuracy.py"""
import torch
torch.manual_seed(0)
x = torch.relu(torch.randn(4096, 256)) * torch.exp(torch.randn(4096, 1) * 2)
w = torch.randn(256, 256) * 0.06
ref = x.double() @ w.double()


def split(t, n, dt, scale=1.0):
    out, r = [], (t * scale).clone()
    for _ in range(n):
        p = r.to(dt).float(); out.append(p); r = r - p
    return out


def prod(pairs, sc=1.0):
    acc = torch.zeros_like(ref)
    for a, b in pairs:
        acc += a.double() @ b.double()
    return acc / sc


res = {"fp32 matmul (fp32 accumulation)": (x @ w).double()}
xs, ws = split(x, 2, torch.bfloat16), split(w, 2, torch.bfloat16)
res["bf16, 2 pieces, 3 products (shipped)"] = prod([(xs[0], ws[0]), (xs[0], ws[1]), (xs[1], ws[0])])
xs, ws = split(x, 3, torch.bfloat16), split(w, 3, torch.bfloat16)
res["bf16, 3 pieces, 6 products"] = prod([(xs[0], ws[0]), (xs[0], ws[1]), (xs[1], ws[0]), (xs[0], ws[2]), (xs[2], ws[0]), (xs[1], ws[1])])
for sx, sw in ((1.0, 1.0), (1.0, 256.0), (16.0, 256.0)):
    xs, ws = split(x, 2, torch.float16, sx), split(w, 2, torch.float16, sw)
    res[f"fp16, 2 pieces, 3 products, x * {sx:g}, w * {sw:g}"] = prod([(xs[0], ws[0]), (xs[0], ws[1]), (xs[1], ws[0])], sx * sw)
row = ref.abs().amax(1, keepdim=True)
for n, v in res.items():
    e = (v - ref).abs()
    print(f"{n:48s} rms relative {float((e ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()):.2e}   max / max|y| {float(e.max() / ref.abs().max()):.2e}   "
          f"worst relative to its own row's largest output {float((e / row).max()):.2e}")
