"""NeRF-subnetwork training step, one process per GPU.

The reference trains with ``torch.nn.DataParallel`` over one process (src/trainer_unified_nuscenes.py:227): every
iteration scatters the batch, replicates the module, gathers the per-replica losses and calls
``loss_total.mean().backward()`` (:334).  Here each rank owns one MI355X and its slice of the batch; the decoder
forward/backward is the HIP training path (``model.train_decoder_weights = True``: per-layer activations / gradients are
written by the kernels, dW on the library BLAS), and the only exchange is ONE all-reduce per step over a flat gradient
bucket (RCCL over xGMI; decoder weights ~1-4 MB plus the code tables), which reproduces the mean-over-replicas gradient.

Functions mirror the NeRF half of ``ParallelModel.forward`` (:117-148) and of ``training_epoch`` (:259-344); the image
encoder / pose-refinement half of that forward is stock PyTorch and outside this package.
"""
from typing import Dict, Iterable, Optional

import torch
import torch.nn as nn


def nerf_losses(model, xyz_batch, viewdir_batch, shapecode_batch, texturecode_batch, z_vals_batch, rgb_tgt_batch,
                occ_pixels_batch, loss_occ_coef: float, composite=None):
    """losses_all, loss_total of the NeRF subnetwork (src/trainer_unified_nuscenes.py:117-148).

    xyz_batch / viewdir_batch (B, n, S, 3); z_vals_batch (B, S); rgb_tgt_batch (B, n, 3); occ_pixels_batch (B, n, 1)."""
    if composite is None:
        from .utils import volume_rendering_batch as composite     # HIP composite kernel; raises on CPU tensors
    sigmas, rgbs = model(xyz_batch.flatten(0, 1), viewdir_batch.flatten(0, 1), shapecode_batch, texturecode_batch)
    b_size = xyz_batch.shape[0]
    n, s, _ = sigmas.shape
    rgb_rays, depth_rays, acc_trans_rays = composite(sigmas.view(b_size, n // b_size, s, -1),
                                                     rgbs.view(b_size, n // b_size, s, -1), z_vals_batch)
    a = torch.abs(occ_pixels_batch)
    denom = a.sum(dim=[-2, -1]) + 1e-9
    loss_rgb = ((rgb_rays - rgb_tgt_batch) ** 2 * a).sum(dim=[-2, -1]) / denom
    loss_occ = (torch.exp(-occ_pixels_batch * (0.5 - acc_trans_rays.unsqueeze(-1))) * a).sum(dim=[-2, -1]) / denom
    loss_reg = torch.norm(shapecode_batch, dim=-1) + torch.norm(texturecode_batch, dim=-1)
    losses_all = {"loss_rgb": loss_rgb.mean(), "loss_occ": loss_occ.mean(), "loss_reg": loss_reg.mean()}
    losses_all["psnr"] = (-10.0 * torch.log(loss_rgb.mean()) / torch.log(torch.tensor(10.0))).detach()
    loss_total = losses_all["loss_rgb"] + loss_occ_coef * losses_all["loss_occ"]
    losses_all["loss_total"] = loss_total
    return losses_all, loss_total


class CodeTables(nn.Module):
    """``shape_codes`` / ``texture_codes`` (src/trainer_unified_nuscenes.py:436-452): one row per object instance,
    randn / sqrt(dim/2) like the reference's ``make_codes``, or every row = the given mean codes; replicated on every rank
    (same seed) and kept in sync by the gradient bucket."""

    def __init__(self, n_objects: int, dim: int = 256, seed: Optional[int] = None, mean_shape=None, mean_texture=None):
        super().__init__()
        self.shape_codes = nn.Embedding(n_objects, dim)
        self.texture_codes = nn.Embedding(n_objects, dim)
        g = None if seed is None else torch.Generator().manual_seed(seed)
        std = 1.0 / (dim / 2) ** 0.5
        with torch.no_grad():
            self.shape_codes.weight.copy_(torch.randn(n_objects, dim, generator=g) * std)
            self.texture_codes.weight.copy_(torch.randn(n_objects, dim, generator=g) * std)
            if mean_shape is not None:
                self.shape_codes.weight.copy_(mean_shape.reshape(1, dim).repeat(n_objects, 1))
                self.texture_codes.weight.copy_(mean_texture.reshape(1, dim).repeat(n_objects, 1))

    def forward(self, idx: torch.Tensor):
        return self.shape_codes(idx), self.texture_codes(idx)


class GradBucket:
    """All trainable gradients as views into ONE flat fp32 buffer, so a step costs one all-reduce.

    ``optimizer.zero_grad(set_to_none=True)`` would drop the views: use ``bucket.zero()`` instead."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("GradBucket: parameters must be fp32 and live on one device")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.group = group
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def check_views(self):
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                raise RuntimeError("GradBucket: a .grad was replaced (zero_grad(set_to_none=True)?); call bucket.zero() instead")

    def allreduce_mean(self):
        """Sum over ranks / world size == gradient of the mean of the per-rank losses (:334)."""
        import torch.distributed as dist
        self.check_views()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))


def learning_rates(hpams: dict, niter: int):
    """src/trainer_unified_nuscenes.py:424-430: both rates halve every ``interval`` iterations."""
    model_lr, latent_lr = hpams["lr_schedule"][0], hpams["lr_schedule"][1]
    return (model_lr["lr"] * 2 ** (-(niter // model_lr["interval"])),
            latent_lr["lr"] * 2 ** (-(niter // latent_lr["interval"])))


def make_optimizer(model: nn.Module, codes: CodeTables, hpams: dict, niter: int = 0):
    """AdamW over model / shape codes / texture codes (src/trainer_unified_nuscenes.py:414-422)."""
    lr1, lr2 = learning_rates(hpams, niter)
    return torch.optim.AdamW([{"params": model.parameters(), "lr": lr1},
                              {"params": codes.shape_codes.parameters(), "lr": lr2},
                              {"params": codes.texture_codes.parameters(), "lr": lr2}])


def train_step(model, codes: CodeTables, opt, bucket: GradBucket, batch: Dict[str, torch.Tensor], loss_occ_coef: float,
               composite=None):
    """One iteration of ``training_epoch`` (src/trainer_unified_nuscenes.py:259-344), NeRF subnetwork only, on this
    rank's slice of the batch.  ``batch``: code_idx (B,), xyz, viewdir, z_vals, rgb_tgt, occ_pixels (already on the
    device, as ``prepare_pixel_samples`` produced them)."""
    sc, tc = codes(batch["code_idx"])
    losses_all, loss_total = nerf_losses(model, batch["xyz"], batch["viewdir"], sc, tc, batch["z_vals"], batch["rgb_tgt"],
                                         batch["occ_pixels"], loss_occ_coef, composite)
    loss_total.backward()
    bucket.allreduce_mean()
    opt.step()
    bucket.zero()
    return {k: v.detach() for k, v in losses_all.items()}
