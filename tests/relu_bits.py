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
