import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import supnerf_amd as A
from supnerf_amd import ops, synthetic as SY
dev = torch.device("cuda:0")
params = {k: v.to(dev) for k, v in SY.init_decoder_params().items()}
pk = ops.pack_weights(params, 3, 1)
P, B = 64, 1
g = torch.Generator().manual_seed(1)
xyz = (torch.rand(P, 3, generator=g) - 0.5).to(dev); vd = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
lat = (torch.randn(B, 4, 256, generator=g) * 0.3).to(dev)
d_sig = torch.randn(P, generator=g).to(dev); d_rgb = torch.randn(P, 3, generator=g).to(dev)
sig, rgb, masks = ops.decoder_fwd(xyz, vd, lat, pk, 3, 1, save_masks=True, precision="fp32")
ref = ops.decoder_bwd(xyz, vd, lat, pk, masks, sig, d_sig, d_rgb, 3, 1, precision="fp32")
got = ops.decoder_bwd(xyz, vd, lat, pk, masks, sig, d_sig, d_rgb, 3, 1, precision="bf16x3")
torch.set_printoptions(linewidth=200, precision=3)
for name, r, q in zip(("d_latent", "d_xyz", "d_dir"), ref, got):
    print(name, "max rel", float((r - q).abs().max() / r.abs().max()))
print("d_xyz per point rel err:", ((ref[1] - got[1]).abs().amax(1) / ref[1].abs().amax()).cpu())
print("d_dir per point rel err:", ((ref[2] - got[2]).abs().amax(1) / ref[2].abs().amax()).cpu())
print("d_lat per layer rel err:", ((ref[0] - got[0]).abs().amax(2) / ref[0].abs().amax()).cpu())
e = (ref[0] - got[0]).abs()[0] / ref[0].abs().max()
print("d_lat layer 3 err by feature block of 16:", e[3].view(16, 16).amax(1).cpu())
print("d_lat layer 0 err by feature block of 16:", e[0].view(16, 16).amax(1).cpu())
# only the colour path: zero d_sig
got2 = ops.decoder_bwd(xyz, vd, lat, pk, masks, sig, torch.zeros_like(d_sig), d_rgb, 3, 1, precision="bf16x3")
ref2 = ops.decoder_bwd(xyz, vd, lat, pk, masks, sig, torch.zeros_like(d_sig), d_rgb, 3, 1, precision="fp32")
print("colour path only: d_dir rel", float((ref2[2] - got2[2]).abs().max() / ref2[2].abs().max()), " d_lat per layer", ((ref2[0] - got2[0]).abs().amax(2) / ref2[0].abs().amax()).cpu())
m1 = torch.full_like(masks, 0xFF)
ref3 = ops.decoder_bwd(xyz, vd, lat, pk, m1, sig, d_sig, d_rgb, 3, 1, precision="fp32")
got3 = ops.decoder_bwd(xyz, vd, lat, pk, m1, sig, d_sig, d_rgb, 3, 1, precision="bf16x3")
print("all-ones masks:", [float((r - q).abs().max() / r.abs().max()) for r, q in zip(ref3, got3)])
print("  d_lat per layer (own max):", [float((ref3[0][0, l] - got3[0][0, l]).abs().max() / ref3[0][0, l].abs().max()) for l in range(4)])
print("own-mask run, d_lat per layer (own max):", [float((ref[0][0, l] - got[0][0, l]).abs().max() / ref[0][0, l].abs().max()) for l in range(4)])
G0 = torch.zeros(8, P, 256, device=dev); G1 = torch.zeros(8, P, 256, device=dev)
ops.decoder_bwd(xyz, vd, lat, pk, masks, sig, d_sig, d_rgb, 3, 1, precision="fp32", layer_grads=G0)
ops.decoder_bwd(xyz, vd, lat, pk, masks, sig, d_sig, d_rgb, 3, 1, precision="bf16x3", layer_grads=G1)
for l in range(7, -1, -1):
    w = 128 if l == 7 else 256
    a, b = G0[l, :, :w], G1[l, :, :w]
    e = (a - b).abs() / a.abs().max()
    print(f"layer {l}: rel err max {float(e.max()):.2e}; by point block of 16: {[round(float(e[16*i:16*i+16].max()), 3) for i in range(P // 16)]}; by feature tile of 16: {[round(float(e[:, 16*t:16*t+16].max()), 2) for t in range(w // 16)]}")
l = 6
a, b = G0[l], G1[l]
print("layer 6 point 0, features 0..31 ref:", a[0, :32].cpu())
print("layer 6 point 0, features 0..31 got:", b[0, :32].cpu())
