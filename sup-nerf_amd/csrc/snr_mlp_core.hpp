// Shared pieces of the fused decoder kernels (forward: snr_mlp.hip, backward: snr_mlp_bwd.hip):
// LDS map, weight-chunk LDS-DMA pipeline, the fp32-MFMA chunk product and layer bookkeeping.
#pragma once
#include "snr_device.hpp"

namespace snr {

constexpr int WBUF = K_VIEW_PAD * KC;               // floats per weight buffer (288 rows x 32, swizzled, no padding)
constexpr int PE_ROW = 97;                          // per-point scratch row: 64 xyz features + 32 dir features + 1
constexpr int PE_WAVE = 32 * PE_ROW;
constexpr int LDS_SCRATCH = 2 * WBUF;               // 4 waves x PE_WAVE
constexpr int COMP_STRIDE = 8;                      // sigma r g b zc + pad
constexpr int LDS_COMP = LDS_SCRATCH + 4 * PE_WAVE;
constexpr int LDS_BIAS = LDS_COMP + 128 * COMP_STRIDE;    // forward: every MFMA layer's bias (n_mfma_layers x 256) and, right behind them as in
                                                          // the packed stream, the two small heads (648 floats), staged once per workgroup
constexpr int LDS_LAT = LDS_BIAS + (2 * MAX_BLOCKS + 4 + 3) * 256;     // forward: the workgroup's latent rows (up to LDS_LAT_ROWS x 256) when it has ONE object
constexpr int LDS_LAT_ROWS = 8;
constexpr int LDS_TOTAL = LDS_LAT + LDS_LAT_ROWS * 256;   // floats
static_assert(LDS_TOTAL * 4 <= 160 * 1024, "LDS budget");

// Weight chunks are stored in the packed stream as the exact LDS image (snr_layout.h: 16-byte slot
// c of row n sits at slot c ^ ((n >> 1) & 7)), so staging is a linear LDS-DMA copy: 1 KiB per
// wave-instruction, no VGPRs, no ds_write.  `rows` x 128 B, 256 threads x 16 B per step.
__device__ __forceinline__ void chunk_dma(const float* __restrict__ g, float* lds, int rows, int tid) {
    const int nvec = rows * 8;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i * 256 < nvec) {    // rows is a multiple of 32: uniform over the workgroup
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)(g + (size_t)(i * 256 + tid) * 4), (lptr_t)(lds + (i * 256 + wave * 64) * 4), 16, 0, 0);
        }
    }
}

// one 32-deep k-chunk: acc[t] += W[32t..32t+31][chunk] * b     (NT output tiles).
// aoff[j] = this lane's float offset of 16-byte slot (2j + h) in row (lane & 31), swizzle applied.
// The A fragment of the next 4 MFMAs is fetched before the current 4 are issued.
// ZERO_C: the accumulators start from zero -- the first MFMA of every tile takes the constant 0 as its C operand instead of a zeroed register set
template <int NT, int NA, int T0 = 0, bool ZERO_C = false>
__device__ __forceinline__ void chunk_mma(f32x16 (&acc)[NA], const float (&b)[16], const float* wbuf, const int (&aoff)[4]) {
    // output tiles T0 .. T0+NT-1
    f32x4 a = *reinterpret_cast<const f32x4*>(wbuf + aoff[0] + T0 * 32 * KC);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = T0; t < T0 + NT; ++t) {
            f32x4 an = a;
            if (t + 1 < T0 + NT) an = *reinterpret_cast<const f32x4*>(wbuf + aoff[j] + (t + 1) * 32 * KC);
            else if (j + 1 < 4) an = *reinterpret_cast<const f32x4*>(wbuf + aoff[j + 1] + T0 * 32 * KC);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[4 * j + e], (ZERO_C && j == 0 && e == 0) ? zero : acc[t], 0, 0, 0);
            }
            a = an;
        }
    }
}

struct Pipe {
    const float* next;   // next chunk to fetch from the packed stream
    int cur;             // LDS buffer holding the chunk about to be consumed
    int aoff[4];         // per-lane A-fragment offsets (see chunk_mma)
};

__device__ __forceinline__ void pipe_init(Pipe& p, const float* stream, int lane) {
    p.next = stream;
    p.cur = 0;
    const int n = lane & 31, h = lane >> 5, sw = (n >> 1) & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) p.aoff[j] = n * KC + (((2 * j + h) ^ sw) << 2);
}

// consume the current chunk while the following one (rows_next x 32; 0 = none) lands in the other buffer
template <int NT, int NA, bool ZERO_C = false>
__device__ __forceinline__ void step(f32x16 (&acc)[NA], const float (&b)[16], Pipe& p, float* lds, int rows_next, int tid,
                                     bool extra_tile = false) {
    if (rows_next) chunk_dma(p.next, lds + (p.cur ^ 1) * WBUF, rows_next, tid);
    chunk_mma<NT, NA, 0, ZERO_C>(acc, b, lds + p.cur * WBUF, p.aoff);
    if (NA > NT && extra_tile) chunk_mma<1, NA, (NA > NT ? NT : 0), ZERO_C>(acc, b, lds + p.cur * WBUF, p.aoff);   // tile NT (backward of enc_viewdir)
    p.next += rows_next * KC;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    p.cur ^= 1;
}

template <int NT, int NA>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[NA], const float* __restrict__ bias, int h) {
    // the loads of tile t+1 are issued before tile t is handed to the accumulators, and that order is pinned: left alone the compiler
    // sinks every load to its use (the register file is full here) and each of the 32 round trips is waited for on its own
    f32x4 cur[4], nxt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[j] = *reinterpret_cast<const f32x4*>(bias + 8 * j + 4 * h);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const f32x4*>(bias + 32 * (t + 1) + 8 * j + 4 * h);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * j + e] = cur[j][e];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
}

// accumulators -> next layer's operand registers: optional ReLU (+ mask bits), optional latent add.  This runs between two layers with the
// matrix pipe idle (hiding it does not pay: tools/_diag/experiments/README.md), so it is kept to 4-5 VALU instructions per value: the ReLU bit
// is the sign of (0 - v), shifted in by one v_alignbit (highest element first, so element r lands on bit r); the max is one v_med3 against an
// opaque +inf (a constant would be rewritten as canonicalise + v_max); without ReLU the floor is -inf and the bits are not stored.
// MASKS false (round 4: a forward that saves no ReLU bits -- inference, evaluation renders, the bench headline): the two instructions per
// value that collect the bit are not issued.  On this chip that is not just issue slots: the fp32 MFMA executes on the vector FP32 ALUs
// (MI355X_MICROARCH.md: "runs at the f32 VECTOR rate"), so every VALU instruction takes its cycles away from the matrix work whether or not
// it is "hidden" (tools/_diag/f32_waves_bench.hip, tools/_diag/experiments/README.md).
// ZADD false (a layer that no latent term follows: three of the eight boundaries of the shipped decoder): the add is not issued either.
template <int NT, int NA, bool MASKS, bool ZADD>
__device__ __forceinline__ void epilogue_impl(const f32x16 (&acc)[NA], float (&in)[9][16], bool relu, const float* __restrict__ zlat, int h,
                                              uint32_t (&mask)[4]) {
    const float lo = relu ? 0.f : -__builtin_inff();
    float hi = __builtin_inff();
    asm volatile("" : "+v"(hi));
#pragma unroll
    for (int i = 0; i < 4; ++i) mask[i] = 0u;
    // (the latent values of tile t+1 are requested before tile t's arithmetic, order pinned: see acc_init_bias)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 zc[4] = {zero4, zero4, zero4, zero4}, zn[4] = {zero4, zero4, zero4, zero4};
    if (ZADD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) zc[j] = *reinterpret_cast<const f32x4*>(zlat + 8 * j + 4 * h);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (ZADD && t + 1 < NT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) zn[j] = *reinterpret_cast<const f32x4*>(zlat + 32 * (t + 1) + 8 * j + 4 * h);
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t m16 = 0u;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
#pragma unroll
            for (int e = 3; e >= 0; --e) {
                float v = acc[t][4 * j + e];
                if (MASKS) m16 = __builtin_amdgcn_alignbit(m16, __float_as_uint(0.f - v), 31);
                v = __builtin_amdgcn_fmed3f(v, lo, hi);
                in[t][4 * j + e] = ZADD ? v + zc[j][e] : v;
            }
        }
        mask[t >> 1] = (t & 1) ? (mask[t >> 1] | (m16 << 16)) : m16;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) zc[j] = zn[j];
    }
}

template <int NT, int NA, bool MASKS = true>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[NA], float (&in)[9][16], bool relu, const float* __restrict__ zlat, int h,
                                         uint32_t (&mask)[4]) {
    if (zlat) epilogue_impl<NT, NA, MASKS, true>(acc, in, relu, zlat, h, mask);          // (uniform)
    else epilogue_impl<NT, NA, MASKS, false>(acc, in, relu, zlat, h, mask);
}

// which latent term (index into the (B,NLAT,256) table) is added after MFMA layer li; -1 = none
__device__ __forceinline__ int latent_after(int li, int sb, int tb) {
    if (li < sb) return li;                          // enc_xyz -> z_1, shape_j -> z_{j+1}
    if (li == sb + 2 && tb > 0) return sb;           // enc_viewdir -> first texture latent
    if (li > sb + 2 && li < sb + 2 + tb) return li - 2;   // texture_j -> next texture latent
    return -1;
}
// index of MFMA layer li among the ReLU layers (enc_shape has none)
__device__ __forceinline__ int relu_slot(int li, int sb) { return li <= sb ? li : li - 1; }

struct DecoderIO {
    const float* packed;
    const float* latent;      // (B, n_lat, 256)
    int sb, tb;
    long long n_points;
    long long points_per_obj;
    float* sigmas;            // (P) optional
    float* rgbs;              // (P,3) optional
    uint4* masks;             // optional
    float* act;               // optional, training: inputs of MFMA layers 1..NL-1 and of rgb.2, [slot][P][256]
    bool live;                // this lane's point exists (set per lane by the kernel)
    const float* latent_bias; // optional (B, n_lat, 256): see snr_render_args::latent_bias (split-bf16 fused forward only)
};

// registers (operand layout) -> row-major [P][256] dump of NT*32 features of one point
template <int NT>
__device__ __forceinline__ void dump_operand(const float (&in)[9][16], float* __restrict__ row /* base of this point's 256 floats */, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = {in[t][4 * j], in[t][4 * j + 1], in[t][4 * j + 2], in[t][4 * j + 3]};
            *reinterpret_cast<f32x4*>(row + 32 * t + 8 * j + 4 * h) = v;
        }
}

// The same dump through the wave's (idle) positional-encoding scratch in LDS: a tile (32 points x 32 features) is written as the lanes hold
// it and read back so that eight consecutive lanes cover one point's 128 bytes -- every store instruction then writes 8 whole cache
// lines instead of 32 quarter lines.  Called by every lane of the wave (rows past the end are skipped per lane); `rows` = the dump rows of
// the wave's 32 points (row r = point tile_first + r), `n_rows` = how many of them exist.  LDS operations of one wave execute in order.
constexpr int DUMP_ROW = 36;      // 32 floats + 4 of padding: conflict-free 16-byte writes (row = lane & 31) and reads (row = lane >> 3)
template <int NT>
__device__ __forceinline__ void dump_operand_staged(const float (&in)[9][16], float* __restrict__ rows, int n_rows, float* scr, int lane) {
    const int p = lane & 31, h = lane >> 5, rr = lane >> 3, cc = lane & 7;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = {in[t][4 * j], in[t][4 * j + 1], in[t][4 * j + 2], in[t][4 * j + 3]};
            *reinterpret_cast<f32x4*>(scr + p * DUMP_ROW + 8 * j + 4 * h) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(scr + r * DUMP_ROW + 4 * cc);
            if (r < n_rows) *reinterpret_cast<f32x4*>(rows + (long long)r * 256 + 32 * t + 4 * cc) = v;
        }
    }
}

struct BwdIO {
    const float* packed;
    const float* latent;
    int sb, tb;
    long long n_points;
    long long points_per_obj;
    const uint4* masks;
    const float* sigmas;     // (P) saved by the forward
    const float* rgbs;       // (P,3) saved by the forward (render mode)
    const float* d_sigmas;   // (P)   upstream, points mode
    const float* d_rgbs;     // (P,3) upstream, points mode
    const float* d_rgb;      // (N,3) upstream, render mode (nullable)
    const float* d_depth;    // (N)
    const float* d_acc;      // (N)
    float* partial;          // [tiles32][n_lat][256] or null
    float* d_xyz;            // (P,3) points mode, nullable
    float* d_dir;            // (P,3) points mode, nullable
    float* d_rays_o;         // (N,3) render mode, nullable
    float* d_rays_d;         // (N,3)
    float* d_t;              // (N,S) per-ray depths only
    float* gdump;            // optional, training: gradient wrt the pre-activation of every MFMA layer, [layer][P][256]
};

}  // namespace snr
