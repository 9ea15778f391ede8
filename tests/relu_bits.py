"""Test helper: the ReLU bits a forward launch saved for its backward (include/supnerf_hip.h: ``relu_masks``), decoded into one boolean
(P, width) tensor per ReLU layer in forward order, for ``oracle.decoder_forward(..., relu_masks=...)``.

Layout (csrc/snr_mlp.hip, csrc/snr_bf16.hip -- both arithmetics store the same): per 32-point wave tile and ReLU layer, 64 lanes x uint4;
lane = 32 h + p holds point 32 tile + p; bit (T & 1) * 16 + r of word T >> 1 is accumulator tile T, register r, i.e. hidden unit
32 T + 8 (r >> 2) + 4 h + (r & 3).  ReLU layers in order: encoding_xyz, shape_layer_1.., encoding_viewdir, texture_layer_1.., rgb.0 (128 units)."""
import torch


def decode_relu_bits(masks_u8: torch.Tensor, n_points: int, shape_blocks: int, texture_blocks: int):
    n_relu = shape_blocks + texture_blocks + 3
    tiles = (n_points + 31) // 32
    words = masks_u8.detach().cpu().contiguous().view(torch.int32)[: tiles * n_relu * 64 * 4].reshape(tiles, n_relu, 2, 32, 4)   # [tile][slot][h][p][word]
    bits = ((words.unsqueeze(-1) >> torch.arange(32, dtype=torch.int32)) & 1).bool().reshape(tiles, n_relu, 2, 32, 128)        # [..][32 word + bit]
    wb = torch.arange(128)
    T, r = 2 * (wb // 32) + ((wb % 32) >> 4), wb % 16
    out = torch.zeros(tiles, n_relu, 32, 256, dtype=torch.bool)
    for h in range(2):
        out[:, :, :, 32 * T + 8 * (r >> 2) + 4 * h + (r & 3)] = bits[:, :, h]
    out = out.permute(1, 0, 2, 3).reshape(n_relu, tiles * 32, 256)[:, :n_points]
    return [out[s] if s < n_relu - 1 else out[s][:, :128] for s in range(n_relu)]


def relu_bits_of(out: torch.Tensor, shape_blocks: int, texture_blocks: int, n_samples: int = 1):
    """The ReLU bits saved by the autograd operator that produced ``out`` (``ops.DecoderPoints`` / ``DecoderPointsTrain``: n_samples = 1;
    ``ops.FusedRender``: the samples per ray), decoded per ReLU layer for the caller's UNPADDED points (the operators pad ragged objects to
    whole wave tiles; the dummy points are dropped here)."""
    node, todo = None, [out.grad_fn]
    while todo:
        f = todo.pop()
        if f is None:
            continue
        if type(f).__name__.startswith(("DecoderPoints", "FusedRender")):
            node = f
            break
        todo.extend(g for g, _ in f.next_functions)
    assert node is not None, "no supnerf_amd operator behind this tensor"
    masks = [t for t in node.saved_tensors if t is not None and t.dtype == torch.uint8]
    assert len(masks) == 1
    B, n, n_pad = node.pad
    per_obj, kept = (n_pad or n) * n_samples, n * n_samples
    layers = decode_relu_bits(masks[0], B * per_obj, shape_blocks, texture_blocks)
    if n_pad:
        layers = [m.reshape(B, per_obj, -1)[:, :kept].reshape(B * kept, -1) for m in layers]
    return layers
