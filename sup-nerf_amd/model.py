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
import importlib
import sys
import threading
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import SnrError


def _leaf(module, name):
    """``module.<name>`` for a parameter WITHOUT ``nn.Module.__getattr__``'s chain of dictionary probes: ``_parameters[name]`` -- or, in a
    replica made by ``torch.nn.parallel.replicate`` (what ``nn.DataParallel`` runs, src/trainer_unified_nuscenes.py:227-229), the plain
    attribute of the same name: replicas carry their (non-leaf, broadcast) parameter copies in ``__dict__``, not in ``_parameters``."""
    p = module._parameters.get(name)
    if p is None:
        p = module.__dict__.get(name)
        if p is None:
            p = getattr(module, name)
    return p


# One lock for the two per-module weight caches (packed stream, stacked latent layers): the decoder is entered from one thread per GPU under
# nn.DataParallel (src/trainer_unified_nuscenes.py:227-229) and a lock on the module itself would not survive ``replicate`` / deepcopy.
_CACHE_LOCK = threading.RLock()

# the arithmetic a newly built decoder starts in (``model.precision``); ``python -m supnerf_amd.run --precision fp32 script.py`` sets it
# for a caller whose code never mentions the attribute
DEFAULT_PRECISION = "auto"


class _DecoderBase(nn.Module):
    """Layer set of src/model_supnerf.py:184-199 and the HIP forward."""

    def _build_decoder(self, shape_blocks, texture_blocks, W, num_xyz_freq, num_dir_freq, latent_dim):
        self._check_shape(shape_blocks, texture_blocks, W, num_xyz_freq, num_dir_freq, latent_dim)
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
        self._init_hip_state()

    @staticmethod
    def _check_shape(shape_blocks, texture_blocks, W, num_xyz_freq, num_dir_freq, latent_dim):
        if W != 256 or latent_dim != 256 or num_xyz_freq != 10 or num_dir_freq != 4:
            raise SnrError("the gfx950 kernels are built for W=256, latent_dim=256, num_xyz_freq=10, num_dir_freq=4 "
                           "(every shipped SUP-NeRF config); got "
                           f"W={W}, latent_dim={latent_dim}, num_xyz_freq={num_xyz_freq}, num_dir_freq={num_dir_freq}")
        if not (0 <= shape_blocks <= 8 and 0 <= texture_blocks <= 8):
            raise SnrError("shape_blocks / texture_blocks must be in 0..8")

    def _init_hip_state(self):
        """Everything the HIP forward keeps on the module besides the reference's own layers (also called on a reference class that
        ``hip_decoder_class`` grafted the forward onto)."""
        self._packed = None
        self._packed_key = None
        # arithmetic of the per-point GEMMs: "fp32" (exact), "bf16x3" (split 16-bit pieces: fp16 forward, bf16 backward) or "auto" (render / optimise: bf16x3 where the shape
        # allows it; training mode: the same split kernels, see forward)
        self.precision = DEFAULT_PRECISION
        # what the last launch actually ran in, and why (queryable record of "auto"'s decision, see ops.PrecisionRecord)
        self.last_precision = None
        # range guard of "auto" (fp16 pieces saturate at +-65504): verdict per weight version, see _auto_precision
        self._guard = {"key": None, "verdict": None, "calls": 0, "detail": None}
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
        return _leaf(m, leaf)

    def _per_point_params(self):
        return {n: self._param(n) for n in ops.per_point_tensor_names(self.shape_blocks, self.texture_blocks)}

    def packed_weights(self) -> torch.Tensor:
        pp = self._per_point_params()
        key = tuple((p.data_ptr(), p._version) for p in pp.values())      # (a move to another device changes data_ptr)
        dev = next(iter(pp.values())).device
        sk = (str(dev), ops.raw_stream(dev) if dev.type == "cuda" else 0)
        with _CACHE_LOCK:
            # One entry per (device, stream): a buffer packed a moment ago by another thread on ITS stream may not be written yet as far
            # as this thread's stream is concerned (re-entrancy: one thread per GPU under nn.DataParallel, or a caller's own thread pool
            # on one module), and a DataParallel replica starts from a shallow copy of its source's attributes on another device.
            cache = self.__dict__.setdefault("_packed_by_stream", {})
            ent = cache.get(sk)
            if ent is None or ent[0] != key:
                while len(cache) >= 8:
                    cache.pop(next(iter(cache)))
                ent = cache[sk] = (key, ops.pack_weights(pp, self.shape_blocks, self.texture_blocks))
            self._packed_key, self._packed = ent             # (the latest, for callers that bind the C ABI themselves: INTEGRATION.md B)
            return ent[1]

    # ---- which arithmetic a call runs in: the recorded, queryable decision of "auto"
    def _auto_precision(self, requested, points_per_obj, probe, train=False):
        """Resolve ``requested`` (the module's ``precision`` or a per-call override) for one forward / backward pair and record the
        decision in ``self.last_precision`` = {"requested", "forward", "backward", "reason"}.

        Explicit requests ("fp32", "bf16x3", tuples) pass through.  "auto" picks the split kernels where the library takes the shape
        (<= 4 blocks, whole 32-point tiles per object) -- and, because their forward chain carries fp16 pieces (activations clamp at
        +-65504 inside the ReLU, weights are packed clamped), only after THIS decoder has been seen to stay in range: the first ``auto``
        call of a weight version runs the same batch once more in exact fp32 (``probe(precision)`` -> outputs; 1.75 ms at 4096 x 64),
        compares, counts clamped weights, and on disagreement beyond ops.RANGE_TOL downgrades the model to fp32 with a warning.  One host
        read per weight version; nothing in the steady state.  Training (weights change every step) and DataParallel replicas (re-made
        every step) re-check on calls 1, 16, 256 and every 1024th (weights drift slowly; a probe costs one exact-fp32 forward + a host read)."""
        sb, tb = self.shape_blocks, self.texture_blocks
        if requested not in (None, "auto"):
            f, b = ops.precision_pair(requested, sb, tb, points_per_obj)
            self.last_precision = {"requested": requested, "forward": f, "backward": b, "reason": "requested"}
            return requested
        if not ops.split_supported(sb, tb, points_per_obj):
            ops.resolve_precision("auto", sb, tb, points_per_obj)          # (says so once per configuration)
            self.last_precision = {"requested": "auto", "forward": "fp32", "backward": "fp32",
                                   "reason": f"shape: the split kernels need shape_blocks + texture_blocks <= 4 and whole 32-point tiles per object "
                                             f"(got {sb} + {tb} blocks, {points_per_obj} points per object)"}
            return "fp32"
        with _CACHE_LOCK:
            g = self._guard
            g["calls"] += 1
            n = g["calls"]
            if train or getattr(self, "_is_replica", False):
                due = g["verdict"] is None or n in (16, 256) or n % 1024 == 0
            else:
                due = g["key"] != self._packed_key or g["verdict"] is None
            if due:
                import warnings
                with torch.no_grad():
                    bad = ops.outputs_disagree(probe("bf16x3"), probe("fp32"))
                    clamped = ops.clamped_weight_count(self._per_point_params().values())
                    bad, clamped = [int(v) for v in torch.stack([bad, clamped]).tolist()]           # the one host read
                g["key"], g["detail"] = self._packed_key, {"values_out_of_tolerance": bad, "weights_beyond_fp16_range": clamped}
                if bad or clamped:
                    if g["verdict"] != "fp32":
                        warnings.warn(f"supnerf_amd: precision 'auto' runs this decoder on the exact fp32 kernels: its split-fp16 forward left "
                                      f"the fp32 forward's values ({bad} outputs beyond {ops.RANGE_TOL:g} relative, {clamped} weights beyond "
                                      f"+-{ops.FP16_MAX:g}); activations or weights exceed the fp16 range", RuntimeWarning, stacklevel=4)
                    g["verdict"] = "fp32"
                else:
                    g["verdict"] = "bf16x3"
            verdict, detail = g["verdict"], g["detail"]
        if verdict == "fp32":
            self.last_precision = {"requested": "auto", "forward": "fp32", "backward": "fp32",
                                   "reason": f"range guard: split forward disagreed with the fp32 forward ({detail})"}
            return "fp32"
        self.last_precision = {"requested": "auto", "forward": "bf16x3", "backward": "bf16x3",
                               "reason": f"split kernels: shape supported, range guard passed ({detail})"}
        return "bf16x3"

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
        ps = [q for l in lat + nxt for q in (_leaf(l, "weight"), _leaf(l, "bias"))]
        dev = ps[0].device
        key = tuple((q.data_ptr(), q._version) for q in ps) + (str(dev), ops.raw_stream(dev) if dev.type == "cuda" else 0)
        with _CACHE_LOCK:
            return self._stacked_locked(lat, nxt, ps, key)

    def _stacked_locked(self, lat, nxt, ps, key):
        if getattr(self, "_stack_key", None) != key:          # (one slot, keyed by weights + device + stream like the packed stream)
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
            x3, d3 = xyz.reshape(-1, 3), viewdir.reshape(-1, 3)
            ppo, probe_pts = self._points_shape(x3, d3, lat, pad=lat.shape[1] > 0)
            prec = self.precision
            if prec in (None, "auto"):
                packed = ops._packed_for(ops.per_point_tensor_names(self.shape_blocks, self.texture_blocks), w, self.shape_blocks,
                                         self.texture_blocks)
                first = self._auto_precision("auto", ppo, lambda p: ops.decoder_fwd(*probe_pts(), lat.detach(), packed, self.shape_blocks,
                                                                                    self.texture_blocks, precision=p)[:2], train=True)
                prec = ("auto", "auto", "bf16x3") if first == "bf16x3" else "fp32"
            else:
                self._auto_precision(prec, ppo, None)
            sig, rgb = ops.DecoderPointsTrain.apply(x3, d3, lat, self.shape_blocks, self.texture_blocks, prec, *w)
            return sig.view(*lead, 1), rgb.view(*lead, 3)
        x3, d3 = xyz.reshape(-1, 3), viewdir.reshape(-1, 3)
        packed = self.packed_weights()
        need_lat = torch.is_grad_enabled() and lat.requires_grad and self.shape_blocks + self.texture_blocks > 0
        ppo, probe_pts = self._points_shape(x3, d3, lat, pad=need_lat)
        prec = self._auto_precision(self.precision, ppo, lambda p: ops.decoder_fwd(*probe_pts(), lat.detach(), packed, self.shape_blocks,
                                                                                   self.texture_blocks, precision=p)[:2])
        sig, rgb = ops.DecoderPoints.apply(x3, d3, lat, packed, self.shape_blocks, self.texture_blocks, prec)
        return sig.view(*lead, 1), rgb.view(*lead, 3)

    @staticmethod
    def _points_shape(x3, d3, lat, pad):
        """(points per object as the operators will launch them -- they pad ragged objects to whole 32-point tiles when latent gradients
        are wanted --, a thunk giving the point tensors in that shape for the range guard's probe launches)."""
        B = max(lat.shape[0], 1)
        P = x3.shape[0]
        per_obj = P // B
        n_pad = ops._tile_pad(per_obj) if (pad and P % B == 0 and P > 0) else 0
        if not n_pad:
            return per_obj, lambda: (x3.detach(), d3.detach())
        return n_pad, lambda: (ops._pad_rows(x3.detach(), B, per_obj, n_pad), ops._pad_rows(d3.detach(), B, per_obj, n_pad))

    def fused_render(self, rays_o, rays_d, t_vals, xyz_div, z_scale, shape_latent, texture_latent, cfg: "ops.RenderCfg"):
        """rays -> (rgb (N,3), depth (N,), acc_trans (N,)) in one launch (see ops.FusedRender)."""
        lat = self.latent_terms(shape_latent, texture_latent)
        packed = self.packed_weights()
        cfg.latent_bias = self.latent_biases(lat)
        n, S = cfg.rays_per_obj, cfg.n_samples
        need_lat = torch.is_grad_enabled() and lat.requires_grad and self.shape_blocks + self.texture_blocks > 0
        n_launch = (ops._tile_pad(n, S) or n) if need_lat else n

        def probe(p):
            # the range check renders ONE object of the batch (a 64-object launch costs 106 ms in exact fp32; the decoder is the same for all)
            B = max(lat.shape[0], 1)
            k = n if B > 1 else rays_o.shape[0]
            first = lambda t, rows: None if t is None else t.detach()[:rows]
            ro, rd = first(rays_o, k), first(rays_d, k)
            tv = None if t_vals is None else (t_vals.detach() if cfg.z_mode == ops.Z_SHARED else first(t_vals, 1 if cfg.z_mode == ops.Z_PER_OBJECT else k))
            per_obj = lambda t: None if t is None else (first(t, 1) if t.shape[0] == B else first(t, k))
            div1, zs, lat1 = per_obj(xyz_div), per_obj(z_scale), lat.detach()[:1] if B > 1 else lat.detach()
            c = ops.copy.copy(cfg)
            if c.latent_bias is not None and B > 1:
                c.latent_bias = c.latent_bias[:1]
            if c.box_half is not None and B > 1:
                c.box_half = c.box_half[:1]
            if n_launch != n:               # ragged objects: padded with dummy rays exactly like FusedRender will (their outputs are compared too)
                if tv is None and c.z_mode == ops.Z_BOX and c.rng is None:
                    tv = torch.zeros(ro.shape[0], S, device=ro.device)          # (any jitter serves a range check; the generator is not touched)
                ro, rd, tv, zs, c = ops.pad_render_inputs(ops._f32c(ro), ops._f32c(rd), ops._f32c(tv), ops._f32c(zs), c, 1, n, n_launch)
            return ops.render_probe(ro, rd, tv, div1, zs, lat1, packed, c, p)

        cfg.precision = self._auto_precision(self.precision if cfg.precision is None else cfg.precision, n_launch * S, probe)
        return ops.FusedRender.apply(rays_o, rays_d, t_vals, xyz_div, z_scale, lat, packed, cfg)


class CodeNeRF(_DecoderBase):
    """Drop-in for src/model_codenerf.py:13 (same constructor defaults)."""

    def __init__(self, shape_blocks=2, texture_blocks=1, W=256, num_xyz_freq=10, num_dir_freq=4, latent_dim=256):
        super().__init__()
        self._build_decoder(shape_blocks, texture_blocks, W, num_xyz_freq, num_dir_freq, latent_dim)


_REF_MODEL_MODULES = ("model_supnerf", "src.model_supnerf")


def _resolve_ref_encoder(module=None):
    """The caller's own ``ImgEncoder`` and ``BasicBlock`` (src/model_supnerf.py:13,17): looked up at run time in the reference's module --
    ``sys.path.insert(0, 'src')`` is the first thing its entry scripts do (optimize_nuscenes.py:1-3) -- never stored here."""
    tried = []
    for name in ([module] if module is not None else []) + list(_REF_MODEL_MODULES):
        mod = name if not isinstance(name, str) else sys.modules.get(name)
        if mod is None:
            try:
                mod = importlib.import_module(name)
            except Exception as e:                       # noqa: BLE001  (ModuleNotFoundError, or the module's own imports failing)
                tried.append(f"{name}: {type(e).__name__}: {e}")
                continue
        enc = getattr(mod, "ImgEncoder", None)
        # an installed module's SUPNeRF was replaced; its encoder never is.  BasicBlock is torchvision's, imported by that module.
        block = getattr(mod, "BasicBlock", None)
        if enc is not None and block is not None:
            return enc, block
        tried.append(f"{getattr(mod, '__name__', mod)}: no ImgEncoder / BasicBlock in it")
    raise SnrError("SUPNeRF was built without an img_encoder and the reference's own ImgEncoder is not importable (it stays the caller's "
                   "stock PyTorch module: put the reference's src/ on sys.path, or pass img_encoder=...).  Tried: " + "; ".join(tried))


class SUPNeRF(_DecoderBase):
    """Decoder + pose-update head of src/model_supnerf.py:164-269 (same constructor keywords, same state-dict).

    The ResNet image encoder (src/model_supnerf.py:17-152) is outside this package and stays the caller's stock PyTorch module:
    ``SUPNeRF(**hpams['net_hyperparams'])`` -- the call of src/optimizer_nuscenes.py:1785 -- resolves the reference's ``ImgEncoder`` from the
    caller's importable ``model_supnerf`` and constructs it exactly as src/model_supnerf.py:168-175 does, so that a strict
    ``load_state_dict(saved['model_params'])`` (:1796) finds every ``img_encoder.*`` key.  ``img_encoder=<module>`` supplies one instead;
    ``img_encoder=False`` builds the decoder + pose head alone (``encode_img`` then raises)."""

    _ref_module = None            # supnerf_amd.install binds the class it puts into a reference module to that module

    def __init__(self, shape_blocks=5, texture_blocks=5, pose_blocks=3, regress_blocks=3, latent_dim=256, pose_dim=16,
                 num_xyz_freq=10, num_dir_freq=4, norm_layer_type="BatchNorm2d", pose_shortcut=False, pred_wlh=False,
                 img_encoder: Optional[nn.Module] = None):
        super().__init__()
        self._check_shape(shape_blocks, texture_blocks, latent_dim, num_xyz_freq, num_dir_freq, latent_dim)
        if img_encoder is None:
            enc_cls, block = _resolve_ref_encoder(self._ref_module)
            norm = nn.InstanceNorm2d if norm_layer_type == "InstanceNorm2d" else nn.BatchNorm2d
            self.img_encoder = enc_cls(block, [3, 4, 6, 3], num_classes=latent_dim, norm_layer=norm, pred_wlh=pred_wlh)
        elif img_encoder is not False:
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
            raise SnrError("this SUPNeRF was built with img_encoder=False (the ResNet encoder stays the caller's stock PyTorch module)")
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



# ------------------------------------------------------------------------------------ the HIP forward on a reference class
_GRAFTED = {}


def hip_decoder_class(ref_cls):
    """A subclass of one of the REFERENCE's own decoder classes -- ``SUPNeRF`` (src/model_supnerf.py:164), ``CodeNeRF``
    (src/model_codenerf.py:13), ``AutoRFMix`` (src/model_autorf.py:190): the same layer names, hence the same ``forward`` text
    (src/model_supnerf.py:241-269) -- whose constructor, state-dict, image encoder and pose head are the reference's and whose
    ``forward`` / ``fused_render`` are this package's.  ``supnerf_amd.install`` puts these into the caller's modules, so that
    ``SUPNeRF(**hpams['net_hyperparams'])`` + strict ``load_state_dict`` + ``encode_img`` + ``pose_update`` + ``nn.DataParallel`` run
    unchanged.  Nothing of the reference is stored: the class is built from the caller's class object at run time."""
    if isinstance(ref_cls, type) and issubclass(ref_cls, _DecoderBase):
        return ref_cls
    hit = _GRAFTED.get(ref_cls)
    if hit is not None:
        return hit

    def __init__(self, *args, **kwargs):
        ref_cls.__init__(self, *args, **kwargs)
        lin = self.encoding_xyz[0]
        W, d_xyz = lin.out_features, lin.in_features
        d_dir = self.encoding_viewdir[0].in_features - W
        n_lat = self.shape_blocks + self.texture_blocks
        latent_dim = getattr(self, "shape_latent_layer_1" if self.shape_blocks else "texture_latent_layer_1")[0].in_features if n_lat else W
        self._check_shape(self.shape_blocks, self.texture_blocks, W, (d_xyz - 3) // 6, (d_dir - 3) // 6, latent_dim)
        self._init_hip_state()

    cls = type(ref_cls.__name__, (_DecoderBase, ref_cls), {
        "__init__": __init__, "__module__": ref_cls.__module__, "__qualname__": ref_cls.__qualname__,
        "__doc__": f"{ref_cls.__module__}.{ref_cls.__name__} with supnerf_amd's HIP forward (supnerf_amd.model.hip_decoder_class).",
        "__supnerf_amd_original__": ref_cls})
    _GRAFTED[ref_cls] = cls
    return cls
