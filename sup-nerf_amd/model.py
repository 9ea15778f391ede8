"""Decoder modules with the reference's constructor / forward signatures, running on the HIP kernels.

``CodeNeRF`` mirrors src/model_codenerf.py:13-63 and ``SUPNeRF`` the decoder + pose-update half of
src/model_supnerf.py:164-269.  Parameter names and shapes are identical to the reference so its
checkpoints (``saved['model_params']``) load with ``load_state_dict``.  The ResNet image encoder of
SUPNeRF is out of scope (it stays a stock PyTorch module supplied by the caller).

What runs where:
  * per-object latent layers ``*_latent_layer_*`` -- stock PyTorch (B x 256 GEMMs; autograd gives the code gradients,
    and the latent-weight gradients in training mode);
  * the per-point decoder (8 GEMMs per sample point) -- ``libsupnerf_hip.so``;
  * ``fused_render`` additionally fuses sampling, frame transforms, positional encoding and the
    alpha composite into the same launch (used by ``supnerf_amd.utils`` / ``supnerf_amd.renderer``).
"""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import SnrError


class _DecoderBase(nn.Module):
    """Layer set of src/model_supnerf.py:184-199 and the HIP forward."""

    def _build_decoder(self, shape_blocks, texture_blocks, W, num_xyz_freq, num_dir_freq, latent_dim):
        if W != 256 or latent_dim != 256 or num_xyz_freq != 10 or num_dir_freq != 4:
            raise SnrError("the gfx950 kernels are built for W=256, latent_dim=256, num_xyz_freq=10, num_dir_freq=4 "
                           "(every shipped SUP-NeRF config); got "
                           f"W={W}, latent_dim={latent_dim}, num_xyz_freq={num_xyz_freq}, num_dir_freq={num_dir_freq}")
        if not (0 <= shape_blocks <= 8 and 0 <= texture_blocks <= 8):
            raise SnrError("shape_blocks / texture_blocks must be in 0..8")
        self.shape_blocks, self.texture_blocks = shape_blocks, texture_blocks
        self.num_xyz_freq, self.num_dir_freq = num_xyz_freq, num_dir_freq
        d_xyz, d_viewdir = 3 + 6 * num_xyz_freq, 3 + 6 * num_dir_freq
        self.encoding_xyz = nn.Sequential(nn.Linear(d_xyz, W), nn.ReLU())
        for j in range(shape_blocks):
            setattr(self, f"shape_latent_layer_{j + 1}", nn.Sequential(nn.Linear(latent_dim, W), nn.ReLU()))
            setattr(self, f"shape_layer_{j + 1}", nn.Sequential(nn.Linear(W, W), nn.ReLU()))
        self.encoding_shape = nn.Linear(W, W)
        self.sigma = nn.Sequential(nn.Linear(W, 1), nn.Softplus())
        self.encoding_viewdir = nn.Sequential(nn.Linear(W + d_viewdir, W), nn.ReLU())
        for j in range(texture_blocks):
            setattr(self, f"texture_latent_layer_{j + 1}", nn.Sequential(nn.Linear(latent_dim, W), nn.ReLU()))
            setattr(self, f"texture_layer_{j + 1}", nn.Sequential(nn.Linear(W, W), nn.ReLU()))
        self.rgb = nn.Sequential(nn.Linear(W, W // 2), nn.ReLU(), nn.Linear(W // 2, 3))
        self._packed = None
        self._packed_key = None
        # arithmetic of the per-point GEMMs: "fp32" (exact), "bf16x3" (split 16-bit pieces: fp16 forward, bf16 backward) or "auto" (render / optimise: bf16x3 where the shape
        # allows it; training mode: the same split kernels, see forward)
        self.precision = "auto"
        # False (optimise / inference, the default): the DECODER is a constant -- codes and poses receive gradients, no decoder weight does
        # (neither the per-point layers nor the per-object latent layers).  That is what the reference's optimisers use: their AdamW
        # holds codes and pose only (src/optimizer_nuscenes.py:1762-1769); the weight gradients torch computes there are never read.
        # True (training): forward() also differentiates every decoder weight, in the arithmetic ``precision`` selects.
        self.train_decoder_weights = False

    # ---- packed per-point weights, re-packed only when a tensor changed
    def _param(self, name):
        """The parameter behind a state-dict key, through the modules' own dictionaries (``dict(self.named_parameters())`` walks the whole
        module tree: ~0.1 ms per call, paid twice per render call)."""
        m = self
        *path, leaf = name.split(".")
        for part in path:
            m = m._modules[part]
        return m._parameters[leaf]

    def _per_point_params(self):
        return {n: self._param(n) for n in ops.per_point_tensor_names(self.shape_blocks, self.texture_blocks)}

    def packed_weights(self) -> torch.Tensor:
        pp = self._per_point_params()
        key = tuple((p.data_ptr(), p._version) for p in pp.values())      # (a move to another device changes data_ptr)
        if self._packed is None or key != self._packed_key:
            self._packed = ops.pack_weights(pp, self.shape_blocks, self.texture_blocks)
            self._packed_key = key
        return self._packed

    # ---- the per-object layers as two GEMMs (one for all latent layers, one for all folded biases) when their weights are constants
    def _latent_params(self):
        mods = self._modules
        lat = [mods[f"shape_latent_layer_{j + 1}"]._modules["0"] for j in range(self.shape_blocks)]
        lat += [mods[f"texture_latent_layer_{j + 1}"]._modules["0"] for j in range(self.texture_blocks)]
        nxt = [mods[f"shape_layer_{j + 1}"]._modules["0"] for j in range(self.shape_blocks)]
        nxt += [mods[f"texture_layer_{j + 1}"]._modules["0"] for j in range(self.texture_blocks)]
        return lat, nxt

    def _stacked(self):
        """(W_lat (512, n_lat*256), b_lat, W_next (n_lat*256, n_lat*256) block diagonal, b_next), rebuilt when a weight changed.
        Row block 0 of W_lat multiplies the shape code, row block 1 the texture code; every latent layer owns one column block."""
        lat, nxt = self._latent_params()
        ps = [q for l in lat + nxt for q in (l._parameters["weight"], l._parameters["bias"])]
        key = tuple((q.data_ptr(), q._version) for q in ps)
        if getattr(self, "_stack_key", None) != key:
            n, W = len(lat), 256
            with torch.no_grad():
                dev = ps[0].device
                w_lat = torch.zeros(2 * W, n * W, device=dev)
                w_nxt = torch.zeros(n * W, n * W, device=dev)
                for j, (l, m) in enumerate(zip(lat, nxt)):
                    r0 = 0 if j < self.shape_blocks else W
                    w_lat[r0:r0 + W, j * W:(j + 1) * W] = l.weight.t()
                    w_nxt[j * W:(j + 1) * W, j * W:(j + 1) * W] = m.weight.t()
                self._stack = (w_lat, torch.cat([l.bias for l in lat]).contiguous(), w_nxt, torch.cat([m.bias for m in nxt]).contiguous())
            self._stack_key = key
        return self._stack

    def latent_terms(self, shape_latent: torch.Tensor, texture_latent: torch.Tensor) -> torch.Tensor:
        """(B, shape_blocks+texture_blocks, 256): z_j = ReLU(Lin_j(code)) (src/model_supnerf.py:253,261), hoisted out
        of the per-ray loop.  With no blocks at all a dummy (B,1,256) of zeros is returned.  When the latent layers' weights are
        constants (``train_decoder_weights`` False, frozen parameters, or no gradient being recorded) all of them are ONE GEMM over
        [shape code | texture code]: two launches forward, two backward (the per-layer form costs ~35 small launches per call with its
        weight gradients); gradients still reach the codes."""
        n_lat = self.shape_blocks + self.texture_blocks
        if n_lat and shape_latent.is_cuda:
            lat_layers, _ = self._latent_params()
            if not self.train_decoder_weights or not torch.is_grad_enabled() or \
                    not any(q.requires_grad for l in lat_layers for q in (l.weight, l.bias)):
                w_lat, b_lat, w_nxt, b_nxt = self._stacked()
                if shape_latent.shape == texture_latent.shape and shape_latent.dim() == 2 and shape_latent.shape[1] == 256 and texture_latent.is_cuda:
                    # one launch for all latent layers AND the biases they fold into (latent_biases hands the second output on)
                    z, lb = ops.LatentLayers.apply(shape_latent, texture_latent, w_lat, b_lat, w_nxt, b_nxt, self.shape_blocks, self.texture_blocks)
                    z._snr_latent_bias = lb
                    return z
                codes = torch.cat([shape_latent, texture_latent], dim=-1)
                return torch.relu(torch.addmm(b_lat, codes, w_lat)).view(shape_latent.shape[0], n_lat, 256)
        outs = [getattr(self, f"shape_latent_layer_{j + 1}")(shape_latent) for j in range(self.shape_blocks)]
        outs += [getattr(self, f"texture_latent_layer_{j + 1}")(texture_latent) for j in range(self.texture_blocks)]
        if not outs:
            return shape_latent.new_zeros(shape_latent.shape[0], 1, 256)
        return torch.stack(outs, dim=1)

    def latent_biases(self, lat: torch.Tensor) -> Optional[torch.Tensor]:
        """(B, shape_blocks+texture_blocks, 256): for every latent term z_j the bias the NEXT layer starts from, b + W z_j.  z_j is
        added after a ReLU (``shape_layer_j(y + z_j)``, src/model_supnerf.py:253-263), so it only reaches that layer through W z_j;
        with these the split-bf16 forward drops the latent add from every epilogue.  No gradient: the backward kernel returns the
        gradient of the latent terms themselves.  None when there are no blocks."""
        if self.shape_blocks + self.texture_blocks == 0:
            return None
        lb = getattr(lat, "_snr_latent_bias", None)          # (computed with the latent terms themselves: ops.LatentLayers)
        if lb is not None and lb.shape == lat.shape:
            return lb
        with torch.no_grad():
            if lat.is_cuda:
                _, _, w_nxt, b_nxt = self._stacked()
                return torch.addmm(b_nxt, lat.detach().reshape(lat.shape[0], -1), w_nxt).view(lat.shape)
            lins = [getattr(self, f"shape_layer_{j + 1}")[0] for j in range(self.shape_blocks)]
            lins += [getattr(self, f"texture_layer_{j + 1}")[0] for j in range(self.texture_blocks)]
            return torch.stack([F.linear(lat[:, j].detach(), lin.weight, lin.bias) for j, lin in enumerate(lins)], dim=1).contiguous()

    def forward(self, xyz, viewdir, shape_latent, texture_latent):
        """sigmas (N,S,1), rgbs (N,S,3) -- same contract as the reference forward
        (src/model_supnerf.py:241-269); rays are object-major over the B codes."""
        if xyz.shape[0] % shape_latent.shape[0]:
            raise SnrError("the number of rays must be divisible by the number of codes (object-major batching)")
        lead = xyz.shape[:-1]
        lat = self.latent_terms(shape_latent, texture_latent)
        if self.train_decoder_weights and torch.is_grad_enabled():
            # training mode (src/trainer_unified_nuscenes.py:120-129,334): gradients also reach the per-point decoder weights
            w = [p for p in self._per_point_params().values()]
            # ``precision`` decides the arithmetic of the step.  Evidence (60 AdamW steps against the same steps on the CPU oracle in float64,
            # tests/test_driver_gpu.py::test_training_outcome_fp32_and_bf16x3_track_the_oracle; all the combinations:
            # tools/_diag/experiments/README.md): WHERE A RUN ENDS UP IS DECIDED BY THE FORWARD CHAIN's arithmetic -- a pre-activation that
            # lands on the other side of zero flips a ReLU and changes the function being differentiated; the backward chain applies the
            # pattern the forward SAVED and has nothing to flip, and the products dW = G^T X do not care either.  With the exact-fp32
            # forward a run ends as far from the float64 run as the reference's own fp32 arithmetic does (loss curve 1.4e-6 - 2.2e-6 against
            # the fp32 oracle's 1.3e-6, weights 1.21e-3 of their movement against 1.21e-3) whatever runs behind it; with a forward on two
            # BF16 pieces per operand (rounds 1-2, 2^-17 per product) it ended 27x / 8x further (3.9e-5, 1.1e-2); with the forward on two
            # FP16 pieces (round 3, 2^-22 per product: the level of fp32 accumulation) it ends at the floor again: 2.1e-6 / 1.21e-3.  So
            #   "auto"   = the split kernels throughout (fp16 pieces forward, bf16 pieces backward and in the products): the reference's
            #              outcome at 0.45 of the exact step's time; exact fp32 where the split kernels do not take the shape,
            #   "fp32"   = exact fp32 throughout (the reference's arithmetic, product for product),
            #   "bf16x3" = the split kernels, insisting (raises where unsupported);
            #   a tuple (forward chain, backward chain[, products]) picks each piece.
            prec = ("auto", "auto", "bf16x3") if self.precision in (None, "auto") else self.precision
            sig, rgb = ops.DecoderPointsTrain.apply(xyz.reshape(-1, 3), viewdir.reshape(-1, 3), lat, self.shape_blocks,
                                                    self.texture_blocks, prec, *w)
            return sig.view(*lead, 1), rgb.view(*lead, 3)
        sig, rgb = ops.DecoderPoints.apply(xyz.reshape(-1, 3), viewdir.reshape(-1, 3), lat, self.packed_weights(),
                                           self.shape_blocks, self.texture_blocks, self.precision)
        return sig.view(*lead, 1), rgb.view(*lead, 3)

    def fused_render(self, rays_o, rays_d, t_vals, xyz_div, z_scale, shape_latent, texture_latent, cfg: "ops.RenderCfg"):
        """rays -> (rgb (N,3), depth (N,), acc_trans (N,)) in one launch (see ops.FusedRender)."""
        lat = self.latent_terms(shape_latent, texture_latent)
        if cfg.precision is None:
            cfg.precision = self.precision
        cfg.latent_bias = self.latent_biases(lat)
        return ops.FusedRender.apply(rays_o, rays_d, t_vals, xyz_div, z_scale, lat, self.packed_weights(), cfg)


class CodeNeRF(_DecoderBase):
    """Drop-in for src/model_codenerf.py:13 (same constructor defaults)."""

    def __init__(self, shape_blocks=2, texture_blocks=1, W=256, num_xyz_freq=10, num_dir_freq=4, latent_dim=256):
        super().__init__()
        self._build_decoder(shape_blocks, texture_blocks, W, num_xyz_freq, num_dir_freq, latent_dim)


class SUPNeRF(_DecoderBase):
    """Decoder + pose-update head of src/model_supnerf.py:164-269 (same constructor keywords).

    ``img_encoder`` is the caller's stock-PyTorch image encoder (the reference's ResNet ``ImgEncoder``,
    src/model_supnerf.py:17-152, is outside this package); ``encode_img`` forwards to it."""

    def __init__(self, shape_blocks=5, texture_blocks=5, pose_blocks=3, regress_blocks=3, latent_dim=256, pose_dim=16,
                 num_xyz_freq=10, num_dir_freq=4, norm_layer_type="BatchNorm2d", pose_shortcut=False, pred_wlh=False,
                 img_encoder: Optional[nn.Module] = None):
        super().__init__()
        if img_encoder is not None:
            self.img_encoder = img_encoder
        W = latent_dim
        self.pose_shortcut, self.pred_wlh = pose_shortcut, pred_wlh
        self._build_decoder(shape_blocks, texture_blocks, W, num_xyz_freq, num_dir_freq, latent_dim)
        self.pose_blocks, self.regress_blocks = pose_blocks, regress_blocks
        self.pose_layer_0 = nn.Sequential(nn.Linear(pose_dim, W), nn.ReLU(inplace=True))
        for j in range(1, pose_blocks):
            setattr(self, f"pose_layer_{j}", nn.Sequential(nn.Linear(W, W), nn.ReLU(inplace=True)))
        self.regress_layer_0 = nn.Sequential(nn.Linear(latent_dim + W, W), nn.ReLU(inplace=True))
        for j in range(1, regress_blocks):
            setattr(self, f"regress_layer_{j}", nn.Sequential(nn.Linear(W, W), nn.ReLU(inplace=True)))
        self.out_delta_layer = nn.Linear(W, 6)

    def encode_img(self, img):
        """src/model_supnerf.py:218-224 (stock PyTorch)."""
        if not hasattr(self, "img_encoder"):
            raise SnrError("SUPNeRF was built without an img_encoder (the ResNet encoder stays a stock PyTorch module)")
        out = self.img_encoder(img, self.pose_shortcut)
        if self.pred_wlh:
            return out
        return (*out, None)

    def pose_update(self, im_feat, box_uv_src):
        """src/model_supnerf.py:226-239 (stock PyTorch)."""
        f = self.pose_layer_0(box_uv_src)
        for j in range(1, self.pose_blocks):
            f = getattr(self, f"pose_layer_{j}")(f)
        d = self.regress_layer_0(torch.cat([im_feat, f], -1))
        for j in range(1, self.regress_blocks):
            d = getattr(self, f"regress_layer_{j}")(d)
        return self.out_delta_layer(d)
