"""NeRF-subnetwork training step, one process per GPU.

The reference trains with ``torch.nn.DataParallel`` over one process (src/trainer_unified_nuscenes.py:227): every
iteration scatters the batch, replicates the module, gathers the per-replica losses and calls
``loss_total.mean().backward()`` (:334).  Here each rank owns one MI355X and its slice of the batch; the decoder
forward/backward is the HIP training path (``model.train_decoder_weights = True``: per-layer activations / gradients are
written by the kernels, the weight gradients come from the library's own split-K MFMA kernels, ``snr_weight_grad``), and the exchange
per step is ONE all-reduce over a flat gradient bucket (RCCL over xGMI; the decoder and latent layers, ~3-16 MB) plus, for the two code
tables, an all-gather of the FEW ROWS each rank touched (B rows of 256 floats per table and rank: at the full nuScenes train split
the tables hold tens of thousands of instances, ~100 MB as a dense bucket, of which a step touches B rows) -- together they reproduce
the mean-over-replicas gradient.  The optimiser still updates every row of the tables like the reference's dense AdamW (weight decay
and the momentum tails of earlier steps act on untouched rows too, src/trainer_unified_nuscenes.py:414-422).

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
    """Gradients of one training step, exchanged the cheap way:

    * dense parameters (decoder, latent layers): views into ONE flat fp32 buffer, so a step costs one all-reduce;
    * ``row_sparse`` parameters (the code tables, (n_instances, 256)): a step touches the few rows of its batch, so every rank keeps the
      table's dense ``.grad`` (zeros elsewhere, the optimiser wants it dense) and the ranks exchange ONLY THE TOUCHED ROWS: one all-gather
      of (row index, gradient rows) -- world x B x 256 floats per table instead of an all-reduce over the whole table.

    ``optimizer.zero_grad(set_to_none=True)`` would drop the views: use ``bucket.zero()`` instead (it clears the flat buffer and only
    the rows of the tables that the last step touched)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, row_sparse: Iterable[torch.nn.Parameter] = ()):
        self.rows = [p for p in row_sparse if p.requires_grad]
        sparse_ids = {id(p) for p in self.rows}
        self.params = [p for p in params if p.requires_grad and id(p) not in sparse_ids]
        if not self.params:
            raise ValueError("GradBucket: no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params + self.rows):
            raise ValueError("GradBucket: parameters must be fp32 and live on one device")
        if any(p.dim() != 2 for p in self.rows):
            raise ValueError("GradBucket: row_sparse parameters are 2-D tables")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.group = group
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        for p in self.rows:
            p.grad = torch.zeros_like(p)
        self._touched = None

    def zero(self):
        self.flat.zero_()
        for p in self.rows:
            if self._touched is None:
                p.grad.zero_()
            else:
                p.grad[self._touched] = 0
        self._touched = None

    def check_views(self):
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                raise RuntimeError("GradBucket: a .grad was replaced (zero_grad(set_to_none=True)?); call bucket.zero() instead")

    def allreduce_mean(self, rows: Optional[torch.Tensor] = None):
        """Sum over ranks / world size == gradient of the mean of the per-rank losses (:334).  ``rows``: the table rows this rank's
        batch touched (``batch["code_idx"]``; needed when the bucket has row-sparse tables)."""
        import torch.distributed as dist
        self.check_views()
        if self.rows and rows is None:
            raise ValueError("GradBucket.allreduce_mean: row-sparse tables need the rows of this step (batch['code_idx'])")
        if self.rows:
            self._touched = rows.to(self.flat.device).long()          # (one rank: duplicates are harmless for the zeroing, and torch.unique would sync)
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return
        world = dist.get_world_size(self.group)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(world)
        if not self.rows:
            return
        # Every rank sends the rows its batch touched -- B index slots and B gradient rows, B = this rank's batch size, known on the HOST
        # (the ranks of a step hold equal slices) -- in one all-gather of the indices and one of the [table 0 | table 1 | ...] rows.  No
        # device -> host read anywhere: a row touched twice holds the sum already and must travel once, so duplicates are blanked (index
        # -1, zero payload) by a sort + first-occurrence mask of fixed length instead of torch.unique (dynamic size = a sync), and the
        # common width is B itself instead of the gathered maximum of the unique counts (round 3: two syncs per step).
        dev = self.flat.device
        t = self._touched
        width = int(t.numel())
        srt, _ = torch.sort(t)
        first = torch.ones_like(srt, dtype=torch.bool)
        first[1:] = srt[1:] != srt[:-1]
        idx = torch.where(first, srt, torch.full_like(srt, -1))
        payload = torch.cat([p.grad[srt] for p in self.rows], dim=1) * first[:, None].to(self.flat.dtype)
        all_idx = [torch.empty_like(idx) for _ in range(world)]
        all_rows = [torch.empty_like(payload) for _ in range(world)]
        dist.all_gather(all_idx, idx, group=self.group)
        dist.all_gather(all_rows, payload, group=self.group)
        # Rebuild the summed rows identically on every rank: one index_add_ PER RANK, in rank order.  Inside one rank's contribution every
        # real index occurs once (the blanked slots point at row 0 with a zero payload: adding an exact zero commutes), so no atomic ever
        # chooses the order of two non-zero addends -- a single index_add_ over the concatenation would (three ranks on one row would give
        # replicas that differ in the last bit and drift apart: nothing re-synchronises the tables).
        off = 0
        for p in self.rows:
            p.grad[t] = 0                                                 # (this rank's own rows come back with everybody's)
            for r in range(world):
                p.grad.index_add_(0, all_idx[r].clamp_min(0), all_rows[r][:, off:off + p.shape[1]] / world)
            off += p.shape[1]
        self._touched = torch.cat(all_idx).clamp_min(0)                    # (for zero(): duplicates and the blanked slots' row 0 are harmless there)
        assert width == all_idx[0].numel()


def learning_rates(hpams: dict, niter: int):
    """src/trainer_unified_nuscenes.py:424-430: both rates halve every ``interval`` iterations."""
    model_lr, latent_lr = hpams["lr_schedule"][0], hpams["lr_schedule"][1]
    return (model_lr["lr"] * 2 ** (-(niter // model_lr["interval"])),
            latent_lr["lr"] * 2 ** (-(niter // latent_lr["interval"])))


def make_optimizer(model: nn.Module, codes: CodeTables, hpams: dict, niter: int = 0):
    """AdamW over model / shape codes / texture codes (src/trainer_unified_nuscenes.py:414-422).  Parameters on the GPU that already
    hold their gradient buffers (a ``GradBucket`` was built first): the same update as ONE launch per step (``ops.TableAdamW``; torch's
    foreach AdamW is ~18 launches = 0.4 ms of a 6.6 ms step; its ``fused=True`` form was tried and does not reproduce the foreach update
    on this ROCm build -- parameters 3.5e-4 apart after three steps at lr 1e-4).  Anything else (CPU tensors, no bucket yet):
    ``torch.optim.AdamW``."""
    lr1, lr2 = learning_rates(hpams, niter)
    groups = [(list(model.parameters()), lr1), (list(codes.shape_codes.parameters()), lr2), (list(codes.texture_codes.parameters()), lr2)]
    trainable = [p for ps, _ in groups for p in ps if p.requires_grad]
    if trainable and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad is not None and p.grad.is_contiguous()
                         and p.grad.shape == p.shape for p in trainable):
        from . import ops
        return ops.TableAdamW(groups)
    return torch.optim.AdamW([{"params": ps, "lr": lr} for ps, lr in groups])


def train_step(model, codes: CodeTables, opt, bucket: GradBucket, batch: Dict[str, torch.Tensor], loss_occ_coef: float,
               composite=None):
    """One iteration of ``training_epoch`` (src/trainer_unified_nuscenes.py:259-344), NeRF subnetwork only, on this
    rank's slice of the batch.  ``batch``: code_idx (B,), xyz, viewdir, z_vals, rgb_tgt, occ_pixels (already on the
    device, as ``prepare_pixel_samples`` produced them)."""
    sc, tc = codes(batch["code_idx"])
    losses_all, loss_total = nerf_losses(model, batch["xyz"], batch["viewdir"], sc, tc, batch["z_vals"], batch["rgb_tgt"],
                                         batch["occ_pixels"], loss_occ_coef, composite)
    loss_total.backward()
    bucket.allreduce_mean(rows=batch["code_idx"] if bucket.rows else None)
    opt.step()
    bucket.zero()
    return {k: v.detach() for k, v in losses_all.items()}
