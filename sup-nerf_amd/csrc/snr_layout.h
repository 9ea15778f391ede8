// Packed decoder-weight layout shared by host launchers and device kernels.
//
// The per-point layers of the decoder (src/model_supnerf.py:184-199 of the reference) are
// consumed by the fused kernels as a linear STREAM of k-chunks: one chunk = ROWS x 32 fp32,
// ROWS = output features of the layer (forward) or input features (backward, transposed
// weights), 32 = a slice of the reduction dimension.  Chunks are stored in exactly the
// order the kernel consumes them so the stream is read front to back with coalesced 16-byte
// LDS-DMA loads.  Reduction dimensions are zero-padded to a multiple of 32.  Inside a chunk the
// 16-byte slots of each row are XOR-swizzled (chunk_pos below).
#pragma once
#include <stdint.h>

#ifndef SNR_HD
#ifdef __HIPCC__
#define SNR_HD __host__ __device__
#else
#define SNR_HD
#endif
#endif

namespace snr {

constexpr int W = 256;           // hidden width == latent width
constexpr int W_RGB = 128;       // rgb.0 output width
constexpr int XYZ_FREQ = 10;
constexpr int DIR_FREQ = 4;
constexpr int D_XYZ = 3 + 6 * XYZ_FREQ;   // 63
constexpr int D_DIR = 3 + 6 * DIR_FREQ;   // 27
constexpr int KC = 32;           // reduction slice per chunk
constexpr int K_XYZ_PAD = 64;    // 63 -> 64
constexpr int K_VIEW_PAD = 288;  // 256 + 27 -> 288
constexpr int MAX_BLOCKS = 8;    // shape_blocks, texture_blocks <= 8

struct Layout {
    int sb, tb;
    int n_mfma_layers;   // sb + tb + 4 : enc_xyz, shape*sb, enc_shape, enc_viewdir, texture*tb, rgb0
    int n_lat;           // sb + tb
    // offsets in floats
    int64_t fwd;         // forward chunk stream
    int64_t bwd;         // backward (transposed) chunk stream
    int64_t bias;        // n_mfma_layers x 256
    int64_t sigma_w;     // 256
    int64_t sigma_b;     // 1 (padded to 4)
    int64_t rgb2_w;      // 3 x 128
    int64_t rgb2_b;      // 3 (padded to 4)
    int64_t bf_fwd;      // split-bf16 forward stream (offset in floats; contents are bf16 pairs)
    int64_t bf_bwd;      // split-bf16 backward stream
    int64_t total;       // floats
    int64_t fwd_floats, bwd_floats;
    int64_t bf_fwd_bytes, bf_bwd_bytes;
};

// ---- split-bf16 ("bf16x3") streams -------------------------------------------------------------------
// Every fp32 weight w is stored as hi = round16(w), lo = round16(w - hi) (fp16 in the forward stream, bf16 in the backward stream).  A chunk
// holds the weights of whole k32-steps as the exact LDS image the kernels read:
//   [k32-step s][16-row tile][plane hi/lo][lane 0..63][8 x 16 bit]          (1 KiB per (tile, plane): lane-linear)
// where element j of lane (n = lane&15, g = lane>>4) is W[row 16*tile + n][k = 32 s + 16 (j>>2) + 4 g + (j&3)] (backward: W^T) --
// the k order in which the 16x16 fp32 accumulator tiles 2s, 2s+1 re-enter v_mfma_f32_16x16x32_* as the B operand.
constexpr int BF_CHUNK = 32768;          // bytes of every chunk except the forward enc_viewdir ones
constexpr int BF_CHUNK_VIEW = 36864;     // 18 k16-steps

SNR_HD inline Layout make_layout(int sb, int tb) {
    Layout L;
    L.sb = sb; L.tb = tb;
    L.n_mfma_layers = sb + tb + 4;
    L.n_lat = sb + tb;
    const int64_t c256 = 256 * KC;
    // forward: enc_xyz 2 chunks, (sb+1) 256-layers x 8, viewdir 9, tb x 8 (all 256 rows), rgb0 8 chunks of 128 rows
    L.fwd_floats = 2 * c256 + (int64_t)(sb + 1) * 8 * c256 + 9 * c256 + (int64_t)tb * 8 * c256 + 8 * (128 * KC);
    // backward: rgb0^T 4 chunks of 256 rows, tb x 8 x 256 rows, viewdir^T 8 chunks of 288 rows,
    //           (sb+1) x 8 x 256 rows, enc_xyz^T 8 chunks of 64 rows
    L.bwd_floats = 4 * c256 + (int64_t)tb * 8 * c256 + 8 * (288 * KC) + (int64_t)(sb + 1) * 8 * c256 + 8 * (64 * KC);
    int64_t o = 0;
    L.fwd = o; o += L.fwd_floats;
    L.bwd = o; o += L.bwd_floats;
    L.bias = o; o += (int64_t)L.n_mfma_layers * 256;
    L.sigma_w = o; o += 256;
    L.sigma_b = o; o += 4;
    L.rgb2_w = o; o += 3 * 128;
    L.rgb2_b = o; o += 4;
    // forward: enc_xyz 2 chunks (4 tiles each, K=64), 8 per 256-layer, 8 x enc_viewdir (K=288), rgb.0 4 chunks
    L.bf_fwd_bytes = 2ll * BF_CHUNK + (int64_t)(sb + 1 + tb) * 8 * BF_CHUNK + 8ll * BF_CHUNK_VIEW + 4ll * BF_CHUNK;
    // backward: rgb.0^T 4 chunks (one k32-step of 16 tiles each, K=128), 8 per 256-layer, enc_viewdir^T 8 + 1 (its two direction tiles), enc_xyz^T 2 (four steps of 4 tiles each)
    L.bf_bwd_bytes = 4ll * BF_CHUNK + (int64_t)tb * 8 * BF_CHUNK + 9ll * BF_CHUNK + (int64_t)(sb + 1) * 8 * BF_CHUNK + 2ll * BF_CHUNK;
    o = (o + 3) & ~3ll;                       // 16-byte alignment for the LDS-DMA source
    L.bf_fwd = o; o += L.bf_fwd_bytes / 4;
    L.bf_bwd = o; o += L.bf_bwd_bytes / 4;
    L.total = o;
    return L;
}

// Position (in floats) of reduction column kk (0..31) of row `row` inside a chunk: the chunk is stored as
// the LDS image the kernels read, i.e. 16-byte slot c = kk/4 of a row sits at slot c ^ ((row >> 1) & 7).
// With 128-byte rows this makes every ds_read_b128 of a 32-row A fragment bank-conflict free (the 16
// lanes of a read group land on 16 distinct 16-byte slots of the 256-byte bank row) and lets the
// staging be a linear LDS-DMA copy.
SNR_HD inline int chunk_pos(int row, int kk) { return row * KC + ((((kk >> 2) ^ ((row >> 1) & 7)) << 2) | (kk & 3)); }

// index of the MFMA layers in consumption order
SNR_HD inline int layer_enc_xyz() { return 0; }
SNR_HD inline int layer_shape(int j /*0-based*/) { return 1 + j; }
SNR_HD inline int layer_enc_shape(int sb) { return 1 + sb; }
SNR_HD inline int layer_viewdir(int sb) { return 2 + sb; }
SNR_HD inline int layer_texture(int sb, int j) { return 3 + sb + j; }
SNR_HD inline int layer_rgb0(int sb, int tb) { return 3 + sb + tb; }

// ReLU masks saved by the forward pass for the backward pass: one bit per hidden unit of every
// ReLU layer, stored per 32-point wave tile as [layer][lane] uint4 (128 bits: the lane's 8 tiles x 16
// accumulator registers).  ReLU layers: enc_xyz, shape*sb, enc_viewdir, texture*tb, rgb0.
SNR_HD inline int n_relu_layers(int sb, int tb) { return sb + tb + 3; }
SNR_HD inline int64_t mask_bytes(int64_t n_points, int sb, int tb) {
    int64_t tiles = (n_points + 31) / 32;
    return tiles * n_relu_layers(sb, tb) * 64 * 16;
}

}  // namespace snr
