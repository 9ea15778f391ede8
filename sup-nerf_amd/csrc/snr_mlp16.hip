// Fused decoder forward for gfx950, exact fp32, TWO waves per SIMD: v_mfma_f32_16x16x4_f32 on 16 points per wave.
//
// Same path as snr_mlp.hip (positional encoding -> code-conditioned MLP on the fp32 MFMA pipe -> optional alpha composite, one
// launch), same packed weight stream, same LDS map, same outputs and ReLU-bit layout.  What differs is the shape of the work:
//
//   * Workgroup = 8 waves = 128 consecutive sample points; wave w owns points 16 w .. 16 w + 15, TWO waves share a SIMD
//     (__launch_bounds__(512, 2): at most 256 registers each).  The matrix pipe is per SIMD: whatever one wave does between its
//     MFMAs -- bias into the accumulators, ReLU / latent add / ReLU bits of the previous layer, the density and colour heads, LDS
//     waits -- runs while the partner wave's MFMAs keep the pipe busy.  With one 512-register wave per SIMD (snr_mlp.hip) that work
//     was exposed: 7.6 % of the kernel at the layer boundaries, which five ways of hiding under the wave's OWN MFMAs did not recover
//     (tools/_diag/experiments/README.md).
//   * GEMMs transposed as before, Y^T = W X^T: the weight slice is the A operand, the wave's 16 points the B / C / D columns.
//     v_mfma_f32_16x16x4_f32 (32 cycles, 64 FLOP/clk/SIMD like 32x32x2): accumulator register r of lane (n = lane & 15, g = lane >> 4)
//     of tile T holds feature 16 T + 4 g + r of point n -- which is the B operand of the NEXT layer's k-slice {16 T + 4 g' + r : g'}
//     when the matching A fragment (row m = lane & 15, k = 16 T + 4 g .. + 3) is fetched with ONE ds_read_b128: the accumulators of
//     layer l are the B operands of layer l + 1 with no data movement, as in the 32x32 kernel.  That fragment is 16 consecutive bytes
//     of the EXISTING packed image (row-major 128-byte rows, 16-byte slots XOR-swizzled by (row >> 1) & 7) and the 64 lanes' reads are
//     bank-conflict free on it (checked for all four ds_read_b128 lane groups), so snr_pack_weights is unchanged.
//   * Layers run k-outer over the previous layer's tiles: input tile T of layer l + 1 is made from accumulator tile T of layer l
//     (ReLU, latent add, ReLU bits, optional activation dump) just before its 64 MFMAs; two accumulator sets (previous layer's
//     values, current sums: 64 + 64 registers), the last k-step deposits the finished tiles in the dead previous set.
//   * One 32 KiB weight chunk (32 k) = two input tiles = 128 MFMAs per wave; ring of two buffers, one barrier per chunk as before.
//
// The fma chain of an output differs from the 32x32x2 kernel's in the ORDER of the k terms (a 16x16x4 MFMA sums k = 16 T + {r, 4 + r,
// 8 + r, 12 + r}, the 32x32x2 form k = 32 c + 8 j + {e, 4 + e}): both are exact fp32 fma chains over the same products, results agree
// to fp32 round-off, not bit for bit.
#include "snr_mlp16_core.hpp"
#include "snr_host.hpp"

namespace snr {

// What happens to a finished accumulator tile of the PREVIOUS layer on its way into the current layer's B operand.
struct Epi {
    float lo;                 // ReLU floor: 0, or -inf for the layer without an activation
    const float* zlds;        // LATLDS: LDS row of the latent term added after the activation (the zero row if none)
    const float* zglb;        // !LATLDS: this lane's global latent row, or null
    float hi;                 // +inf, opaque (made once per kernel): the upper bound of the ReLU's v_med3
    float* dump;              // training: this lane's row of the activation dump ([point][256] + 4 g), or null
};
struct EpiRegs { f32x4 z; };          // what a tile's epilogue reads from LDS, requested a tile ahead

__device__ __forceinline__ uint32_t spread_nibbles(uint32_t m16) {
    // nibble of input tile dT (shifted in first = highest) -> bits 8 dT .. 8 dT + 3
    return ((m16 >> 12) & 0xFu) | (((m16 >> 8) & 0xFu) << 8) | (((m16 >> 4) & 0xFu) << 16) | ((m16 & 0xFu) << 24);
}

template <bool LATLDS>
__device__ __forceinline__ EpiRegs epi_load(const Epi& c, int T, int g) {
    EpiRegs e;
    e.z = f32x4{0.f, 0.f, 0.f, 0.f};
    if (LATLDS) e.z = *reinterpret_cast<const f32x4*>(c.zlds + 16 * T + 4 * g);
    else if (c.zglb) e.z = *reinterpret_cast<const f32x4*>(c.zglb + 16 * T + 4 * g);
    return e;
}

// one value of accumulator tile T of the previous layer -> B-operand register r of input tile T (feature 16 T + 4 g + r): ReLU (or none),
// ReLU bit (shifted in: call with r = 3, 2, 1, 0), the density head's term, the latent add.  Five VALU instructions, placed by the caller
// BETWEEN the MFMA groups of the tile before (see layer_from_acc).
template <bool MASKS>
__device__ __forceinline__ void epi_value(const f32x4& acc, f32x4& x, const Epi& c, const EpiRegs& e, int r, uint32_t& m16) {
    float v = acc[r];
    if (MASKS) m16 = __builtin_amdgcn_alignbit(m16, __float_as_uint(0.f - v), 31);      // bit = (v > 0): the sign of 0 - v (+0 and -0 give 0)
    v = __builtin_amdgcn_fmed3f(v, c.lo, c.hi);
    x[r] = v + e.z[r];
    if (MASKS) asm volatile("" : "+v"(m16));      // (pinned: left alone the compiler sinks the bit collection to the end of the layer)
}

template <int NT>
__device__ __forceinline__ void acc_from_bias(f32x4 (&acc)[16], const float* bias /* LDS, layer's row */, int g) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * g);
}

// One layer whose K = 256 inputs are the previous layer's accumulators accP: NT output tiles (16: 256 rows; 8: rgb.0), chunks
// base + 0 .. 7 of the stream.  `pre` (uniform; enc_viewdir): the layer also has a chunk of explicit operand tiles xe (the direction
// features) -- chunk 8 of its stream, consumed FIRST, so that the last MFMA step of every layer is the same deposit into accP (with the
// extra chunk behind the eight, the merge of the two ends of the layer cost 48 spilled registers).  On entry the layer's first chunk
// (the direction chunk if `pre`) is in the current buffer; at its last chunk the layer requests `next_first` (next_rows rows; 0 = the
// stream ends).  On return accP holds this layer's sums (bias included, no activation yet).
template <int NT, int WAVES, bool LATLDS, bool MASKS, bool DUMP>
__device__ __forceinline__ void layer_from_acc(f32x4 (&accP)[16], Ring16& ring, float* lds, const float* bias, const Epi& c, int g, const Dma16& dm,
                                               const float* base, bool pre, const f32x4 (&xe)[2], const float* next_first, int next_rows,
                                               uint32_t (&mw)[4]) {
    f32x4 accC[16];
    f32x4 a0, a1;
    constexpr int rows_mid = NT * 16;
    constexpr int chunk_floats = rows_mid * KC;
    if (pre) {
        chunk_dma16<WAVES>(dm, base, lds + (ring.cur ^ 1) * WBUF16, lds, rows_mid);
        const float* wb = lds + ring.cur * WBUF16;
        first_pair(a0, a1, wb + ring.aoff[0]);
        acc_from_bias<NT>(accC, bias, g);
        tile_mma<NT, false>(accC, accP, xe[0], wb + ring.aoff[0], a0, a1, wb + ring.aoff[1]);
        tile_mma<NT, false>(accC, accP, xe[1], wb + ring.aoff[1], a0, a1, nullptr);
        ring_turn(ring);
    } else {
        acc_from_bias<NT>(accC, bias, g);
    }
    // Input tile T's four operand registers are made from accP[T] while tile T - 1's MFMAs run: one value (five VALU instructions) behind
    // every second MFMA group of that tile, its latent / head-weight rows requested behind the first group; the LDS-DMA pieces of the next
    // chunk go out one per group as well.  Nothing but the tile's own MFMAs then stands between two tiles -- with the previous form (whole
    // epilogue + four DMA pieces at the top of a tile) both waves of a SIMD, which run this program in lockstep, left the matrix pipe idle
    // together.  Only the layer's first tile is made up front.
    constexpr int NG = NT / 2;                   // MFMA groups per tile
    EpiRegs e = epi_load<LATLDS>(c, 0, g);
    f32x4 xa, xb;
    uint32_t m16 = 0u;
#pragma unroll
    for (int r = 3; r >= 0; --r) epi_value<MASKS>(accP[0], xa, c, e, r, m16);
    if (DUMP) *reinterpret_cast<f32x4*>(c.dump) = xa;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const float* src = (ch < 7) ? base + (ch + 1) * chunk_floats : next_first;
        const int rows = (ch < 7) ? rows_mid : next_rows;
        float* const dst = lds + (ring.cur ^ 1) * WBUF16;
        const float* wb = lds + ring.cur * WBUF16;
        first_pair(a0, a1, wb + ring.aoff[0]);
        // first tile of the chunk (T = 2 ch) on xa; between its groups: the next chunk's DMA pieces and tile 2 ch + 1's operand
        tile_mma<NT, false>(accC, accP, xa, wb + ring.aoff[0], a0, a1, wb + ring.aoff[1], [&](int gi) {
            if (gi == 0) e = epi_load<LATLDS>(c, 2 * ch + 1, g);
            constexpr int PSTEP = (WAVES == 8) ? 2 : 1;      // a piece covers 8 WAVES rows: 4 (8 waves) or 8 (4 waves) pieces per 256-row chunk
            if (gi % PSTEP == 0 && (gi / PSTEP) * 8 * WAVES < rows) chunk_piece16<WAVES>(dm, src, dst, lds, rows, gi / PSTEP);
#pragma unroll
            for (int k = (gi * 4) / NG; k < ((gi + 1) * 4) / NG; ++k) epi_value<MASKS>(accP[2 * ch + 1], xb, c, e, 3 - k, m16);
            if (DUMP && gi == NG - 1) *reinterpret_cast<f32x4*>(c.dump + 16 * (2 * ch + 1)) = xb;
        });
        if (ch & 1) { mw[ch >> 1] = m16; m16 = 0u; }
        // second tile (T = 2 ch + 1) on xb; between its groups: tile 2 ch + 2's operand (the next chunk's first tile)
        if (ch == 7) {
            tile_mma<NT, true>(accC, accP, xb, wb + ring.aoff[1], a0, a1, nullptr);
        } else {
            tile_mma<NT, false>(accC, accP, xb, wb + ring.aoff[1], a0, a1, nullptr, [&](int gi) {
                if (gi == 0) e = epi_load<LATLDS>(c, 2 * ch + 2, g);
#pragma unroll
                for (int k = (gi * 4) / NG; k < ((gi + 1) * 4) / NG; ++k) epi_value<MASKS>(accP[2 * ch + 2], xa, c, e, 3 - k, m16);
                if (DUMP && gi == NG - 1) *reinterpret_cast<f32x4*>(c.dump + 16 * (2 * ch + 2)) = xa;
            });
        }
        ring_turn(ring);
    }
}

// the lane's 64 ReLU bits of a layer (4 words x 16 bits: nibbles of input tiles 4 w .. 4 w + 3) -> the documented layout (snr_layout.h: per
// 32-point tile [layer][lane (p, h)] uint4, word w bit 16 (t & 1) + 4 j + e <-> feature 32 t + 8 j + 4 h + e) and stored.  Feature 16 T + 4 g + r
// is bit 8 (T & 3) + 4 (g >> 1) + r of word T >> 2 of lane (p, h = g & 1): lanes g and g ^ 2 (= lane ^ 32) hold the two nibble columns.
__device__ __forceinline__ void store_masks16x4(uint4* __restrict__ dst /* tile's [64] uint4 of this layer */, const uint32_t (&mw)[4], int wave, int lane) {
    const int n = lane & 15, g = lane >> 4;
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t mine = spread_nibbles(mw[i]) << (4 * (g >> 1));
        w[i] = mine | (uint32_t)__shfl_xor((int)mine, 32, 64);
    }
    if (g < 2) dst[16 * (wave & 1) + n + 32 * g] = make_uint4(w[0], w[1], w[2], w[3]);
}

#ifdef SNR_STAMPS   /* diagnostic build (tools/build_diag.sh): lane 0 of every wave writes s_memtime at phase boundaries into the sigmas buffer, 8 per
                       16-point wave tile (tools/_diag/stamps16.py prints them); the renders of such a build are valid, the saved sigmas are not */
#define SNR16_STAMP(i) do { if (io.sigmas && lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        reinterpret_cast<unsigned long long*>(io.sigmas)[(tile_wg * WAVES + wave) * 8 + (i)] = t_; } } while (0)
/* slot 6 takes the wave's place on the chip instead: HW_ID (wave, SIMD, CU, SE) | XCC_ID << 32 */
#define SNR16_STAMP_HWID(i) do { if (io.sigmas && lane == 0) { unsigned a_, b_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(a_), "=s"(b_)); \
        reinterpret_cast<unsigned long long*>(io.sigmas)[(tile_wg * WAVES + wave) * 8 + (i)] = (unsigned long long)a_ | ((unsigned long long)b_ << 32); } } while (0)
#else
#define SNR16_STAMP(i) do {} while (0)
#define SNR16_STAMP_HWID(i) do {} while (0)
#endif

#ifdef SNR16_NO_PRIO
#define SNR16_PRIO(p) do {} while (0)
#else
#define SNR16_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif

template <int MODE, int WAVES, bool LATLDS, bool MASKS, bool DUMP>
__global__ void __launch_bounds__(WAVES * 64, 2)
decoder_fwd16_kernel(DecoderIO io, Layout L, Lds16 lo, const float* __restrict__ xyz, const float* __restrict__ viewdir, RayGeom gm,
                     float* __restrict__ out_rgb, float* __restrict__ out_depth, float* __restrict__ out_acc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WGP = WAVES * 16;                      // points per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const long long tile_wg = blockIdx.x;
    const long long gp_raw = tile_wg * WGP + wave * 16 + n;
    const bool live = gp_raw < io.n_points;
    const long long gp = live ? gp_raw : io.n_points - 1;
    const long long tile32 = tile_wg * (WAVES / 2) + (wave >> 1);
    const bool tile_live = tile32 * 32 < io.n_points;
    const int sb = io.sb, tb = io.tb;
    const int n_relu = n_relu_layers(sb, tb);
    SNR16_STAMP(0);
    // The prologue and the tail are VALU work; the SIMD's other wave (another workgroup, out of phase) is mid-stream, issuing MFMAs back to back,
    // and at equal priority its ready MFMA wins the ALUs nearly every time: the prologue then takes 69 000 cycles instead of 9 000 and the matrix
    // pipe is fed by one wave (0.87 of its rate) for that long.  With priority the prologue is over in a fraction of that and two waves share
    // the pipe again (tools/_diag/stamps16.py).
    SNR16_PRIO(3);
    float px, py, pz, dx, dy, dz, zc = 0.f;
    if (MODE == 0) {
        px = xyz[gp * 3]; py = xyz[gp * 3 + 1]; pz = xyz[gp * 3 + 2];
        dx = viewdir[gp * 3]; dy = viewdir[gp * 3 + 1]; dz = viewdir[gp * 3 + 2];
    } else {
        const PointId id = point_id(gm, tile_wg, WGP, wave * 16 + n, live);
        const SamplePoint sp = make_sample(gm, id.ray, id.s, id.obj);
        px = sp.x; py = sp.y; pz = sp.z; dx = sp.dx; dy = sp.dy; dz = sp.dz; zc = sp.zc;
    }
    const float* bias = lds + lo.bias;
    const float* heads = bias + L.n_mfma_layers * 256;       // sigma_w (256) | sigma_b | rgb2_w (384) | rgb2_b, as in the packed stream
    const float* zero = lds + lo.zero;
    const float* lat_lane = io.latent + (gp / io.points_per_obj) * (long long)L.n_lat * 256;       // (!LATLDS: the lane's own object)
    const float* lat_wg = io.latent + ((tile_wg * WGP) / io.points_per_obj) * (long long)L.n_lat * 256;

    // ---- prologue: first weight chunk, biases + heads (+ latent rows) by LDS-DMA while the positional encodings are computed
    Ring16 ring;
    ring.cur = 0;
    Dma16 dm;
    dm.voff = lane * 16u + 4096u;
    dm.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds);
    dm.wave = __builtin_amdgcn_readfirstlane(wave);
    {
        const int sw = (n >> 1) & 7;
        ring.aoff[0] = n * KC + (((0 + g) ^ sw) << 2);
        ring.aoff[1] = n * KC + (((4 + g) ^ sw) << 2);
    }
    // the forward stream (snr_layout.h): enc_xyz 2 chunks, then 8 per 256-wide layer (9 for enc_viewdir: the direction chunk is its ninth),
    // rgb.0's 8 chunks of 128 rows.  A layer's FIRST consumed chunk is its chunk 0 -- enc_viewdir's is the direction chunk.
    constexpr long long C256 = 256 * KC;
    const int li_encshape = sb + 1, li_view = sb + 2, li_last = sb + tb + 2;
    const float* const stream = io.packed + L.fwd;
    auto layer_base = [&](int li) { return stream + 2 * C256 + (long long)(li - 1) * 8 * C256 + (li > li_view ? C256 : 0); };      // li = 1 .. li_last + 1
    auto layer_first = [&](int li) { return layer_base(li) + (li == li_view ? 8 * C256 : 0); };
    f32x4 xin[4], xd[2];
    {
        chunk_dma16<WAVES>(dm, stream, lds, lds, 256);
        for (int r = dm.wave; r < L.n_mfma_layers + 3; r += WAVES) row_dma16(dm, io.packed + L.bias + r * 256, lds + lo.bias + r * 256, lds);
        if (LATLDS)
            for (int r = dm.wave; r < L.n_lat; r += WAVES) row_dma16(dm, lat_wg + r * 256, lds + lo.lat + r * 256, lds);
        if (tid < 64) *reinterpret_cast<f32x4*>(lds + lo.zero + 4 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};
        float* sc = lds + lo.scratch + wave * PE_WAVE16 + n * PE_ROW;
        // 30 (frequency, axis) pairs of the xyz encoding, 8 per lane group (the fourth takes 6); 12 of the direction encoding, 3 each
#pragma unroll 1
        for (int i = 0; i < 8; i += 2) {                   // two pairs per trip on packed arithmetic (pe_sincos2: half the VALU instructions)
            const int q = 8 * g + i;
            if (q < 3 * XYZ_FREQ) {
                f32x2 sn, cs;
                pe_sincos2(f32x2{ldexpf(pick3(px, py, pz, q % 3), q / 3), ldexpf(pick3(px, py, pz, (q + 1) % 3), (q + 1) / 3)}, &sn, &cs);
                sc[3 + q] = sn[0]; sc[3 + 3 * XYZ_FREQ + q] = cs[0];
                sc[4 + q] = sn[1]; sc[4 + 3 * XYZ_FREQ + q] = cs[1];
            }
        }
        {
            const int q = 3 * g;
            f32x2 sn, cs;
            pe_sincos2(f32x2{ldexpf(pick3(dx, dy, dz, q % 3), q / 3), ldexpf(pick3(dx, dy, dz, (q + 1) % 3), (q + 1) / 3)}, &sn, &cs);
            sc[64 + 3 + q] = sn[0]; sc[64 + 3 + 3 * DIR_FREQ + q] = cs[0];
            sc[64 + 4 + q] = sn[1]; sc[64 + 4 + 3 * DIR_FREQ + q] = cs[1];
            float s1, c1;
            pe_sincos(ldexpf(pick3(dx, dy, dz, (q + 2) % 3), (q + 2) / 3), &s1, &c1);
            sc[64 + 5 + q] = s1; sc[64 + 5 + 3 * DIR_FREQ + q] = c1;
        }
        if (g == 0) {
            sc[0] = px; sc[1] = py; sc[2] = pz; sc[63] = 0.f;
            sc[64] = dx; sc[65] = dy; sc[66] = dz;
#pragma unroll
            for (int f = D_DIR; f < 32; ++f) sc[64 + f] = 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // B-operand registers: register r of input tile T <- feature 16 T + 4 g + r of the lane's point
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) xin[T][r] = sc[16 * T + 4 * g + r];
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) xd[T][r] = sc[64 + 16 * T + 4 * g + r];
        if (WAVES == 4) __syncthreads();      // the scratch lies over ring buffer 1: every wave has read its operands before any wave requests chunk 1
    }

    SNR16_PRIO(0);
    SNR16_STAMP(1);
    // ---- enc_xyz: 64 -> 256 from explicit operand tiles (two chunks)
    f32x4 accP[16];
    {
        f32x4 accC[16];
        f32x4 a0, a1;
        acc_from_bias<16>(accC, bias, g);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            chunk_dma16<WAVES>(dm, ch == 0 ? stream + C256 : layer_first(1), lds + (ring.cur ^ 1) * WBUF16, lds, 256);
            const float* wb = lds + ring.cur * WBUF16;
            first_pair(a0, a1, wb + ring.aoff[0]);
            tile_mma<16, false>(accC, accP, xin[2 * ch], wb + ring.aoff[0], a0, a1, wb + ring.aoff[1]);
            if (ch == 1) tile_mma<16, true>(accC, accP, xin[2 * ch + 1], wb + ring.aoff[1], a0, a1, nullptr);
            else tile_mma<16, false>(accC, accP, xin[2 * ch + 1], wb + ring.aoff[1], a0, a1, nullptr);
            ring_turn(ring);
        }
    }

    SNR16_STAMP(2);
    // ---- the 256-wide middle layers (shape blocks, enc_shape, enc_viewdir, texture blocks): layer li consumes layer li - 1's sums
    uint32_t mw[4] = {0u, 0u, 0u, 0u};
    float o_sigma = 0.f;
    float pos_inf = __builtin_inff();
    asm volatile("" : "+v"(pos_inf));          // (opaque: a literal +inf in v_med3 would be rewritten as canonicalise + v_max, per value)
    // training dumps: a lane past the end holds the LAST point (gp is clamped), i.e. the same values as that point's own lane: its stores
    // repeat that lane's bytes at that lane's address, so no per-lane predicate (= no divergent branch per tile) is needed
    float* const dump_lane = DUMP ? io.act + gp * 256 + 4 * g : nullptr;
    auto epi_of = [&](int lp) {          // the epilogue applied to the OUTPUT of layer lp on its way into layer lp + 1
        const int la = latent_after(lp, sb, tb);
        Epi c;
        c.lo = (lp != li_encshape) ? 0.f : -__builtin_inff();
        c.zlds = (la >= 0) ? lds + lo.lat + la * 256 : zero;
        c.zglb = (la >= 0) ? lat_lane + la * 256 : nullptr;
        c.hi = pos_inf;
        c.dump = DUMP ? dump_lane + (long long)lp * io.n_points * 256 : nullptr;
        return c;
    };
    auto after_layer_input = [&](int lp) {      // ReLU bits of layer lp, collected while it was consumed
        if (MASKS && lp != li_encshape && tile_live) store_masks16x4(io.masks + (tile32 * n_relu + relu_slot(lp, sb)) * 64, mw, wave, lane);
    };
#pragma unroll 1
    for (int li = 1; li <= li_last; ++li) {
        const int lp = li - 1;
        const Epi c = epi_of(lp);
        if (li == li_view) {
            SNR16_STAMP(3);
            // density head on enc_shape's finished sums (accP, no activation, no latent): softplus(w_sigma . y + b), src/model_supnerf.py:257.
            // One pass of 64 fma here instead of an fma + a head-weight fetch in EVERY layer's per-value epilogue: on this chip a VALU
            // instruction is not hidden by fp32 MFMAs, it competes with them for the same ALUs.  The lane groups hold a quarter of the features.
            float s = 0.f;
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                const f32x4 ws = *reinterpret_cast<const f32x4*>(heads + 16 * T + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fmaf(ws[r], accP[T][r], s);
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float pre = s + heads[L.sigma_b - L.sigma_w];
            o_sigma = pre > 20.f ? pre : log1pf(expf(pre));
        }
        layer_from_acc<16, WAVES, LATLDS, MASKS, DUMP>(accP, ring, lds, bias + li * 256, c, g, dm, layer_base(li), li == li_view, xd, layer_first(li + 1),
                                                       (li == li_last) ? 128 : 256, mw);
        after_layer_input(lp);
    }
    SNR16_STAMP(4);
    {   // rgb.0: 256 -> 128
        const Epi c = epi_of(li_last);
        layer_from_acc<8, WAVES, LATLDS, MASKS, DUMP>(accP, ring, lds, bias + (li_last + 1) * 256, c, g, dm, layer_base(li_last + 1), false, xd, nullptr, 0, mw);
        after_layer_input(li_last);
    }

    SNR16_STAMP(5);
    SNR16_PRIO(3);
    // ---- rgb.0's output (accP tiles 0..7): ReLU + bits, then rgb.2 (128 -> 3) on the VALU
    float o_r, o_g, o_b;
    {
        const float* w2 = heads + (L.rgb2_w - L.sigma_w);
        float pr = 0.f, pg = 0.f, pb = 0.f;
        uint32_t m16 = 0u, mr[4] = {0u, 0u, 0u, 0u};
        float* const dmp = DUMP ? dump_lane + (long long)(li_last + 1) * io.n_points * 256 : nullptr;
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const f32x4 wr = *reinterpret_cast<const f32x4*>(w2 + 16 * T + 4 * g);
            const f32x4 wg = *reinterpret_cast<const f32x4*>(w2 + 128 + 16 * T + 4 * g);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w2 + 256 + 16 * T + 4 * g);
            f32x4 xv;
#pragma unroll
            for (int r = 3; r >= 0; --r) {
                float v = accP[T][r];
                m16 = __builtin_amdgcn_alignbit(m16, __float_as_uint(0.f - v), 31);
                v = fmaxf(v, 0.f);
                xv[r] = v;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr = fmaf(wr[r], xv[r], pr); pg = fmaf(wg[r], xv[r], pg); pb = fmaf(wb[r], xv[r], pb); }
            if (DUMP) *reinterpret_cast<f32x4*>(dmp + 16 * T) = xv;
            if ((T & 3) == 3) { mr[T >> 2] = m16; m16 = 0u; }
        }
        if (MASKS && tile_live) store_masks16x4(io.masks + (tile32 * n_relu + (n_relu - 1)) * 64, mr, wave, lane);
        pr += __shfl_xor(pr, 16, 64); pr += __shfl_xor(pr, 32, 64);
        pg += __shfl_xor(pg, 16, 64); pg += __shfl_xor(pg, 32, 64);
        pb += __shfl_xor(pb, 16, 64); pb += __shfl_xor(pb, 32, 64);
        const float* b2 = heads + (L.rgb2_b - L.sigma_w);
        o_r = pr + b2[0]; o_g = pg + b2[1]; o_b = pb + b2[2];
    }

    SNR16_STAMP_HWID(6);
#ifndef SNR_STAMPS
    if (live && g == 0 && io.sigmas) io.sigmas[gp] = o_sigma;
#endif
    if (live && g == 0) {
        if (io.rgbs) { io.rgbs[gp * 3] = o_r; io.rgbs[gp * 3 + 1] = o_g; io.rgbs[gp * 3 + 2] = o_b; }
    }
    if (MODE == 1) {
        float* comp = lds + lo.comp;             // (WAVES = 4: over ring buffer 0 -- every wave is past the last chunk's barrier)
        if (g == 0) {
            float* c = comp + (wave * 16 + n) * COMP_STRIDE;
            c[0] = o_sigma; c[1] = o_r; c[2] = o_g; c[3] = o_b; c[4] = zc;
        }
        __syncthreads();
        const int S = gm.S;
        const int rays_here = WGP / S;           // host guarantees WGP % S == 0
        const bool white = gm.flags & SNR_WHITE_BKGD;
        for (int r = wave; r < rays_here; r += WAVES) {
            const long long ray = tile_wg * rays_here + r;
            if (ray >= gm.n_rays) break;
            const float* c0 = comp + r * S * COMP_STRIDE;
            RayOut o = composite_ray_fwd(S, lane, white, [&](int k, float& s_, float& r_, float& g_, float& b_, float& z_, float& zn_) {
                const float* c = c0 + k * COMP_STRIDE;
                s_ = c[0]; r_ = c[1]; g_ = c[2]; b_ = c[3]; z_ = c[4];
                zn_ = (k < S - 1) ? c[COMP_STRIDE + 4] : 0.f;
            });
            if (lane == 0) {
                out_rgb[ray * 3] = o.r; out_rgb[ray * 3 + 1] = o.g; out_rgb[ray * 3 + 2] = o.b;
                out_depth[ray] = o.depth; out_acc[ray] = o.acc;
            }
        }
    }
    SNR16_STAMP(7);
}

}  // namespace snr

using namespace snr;

// mode 0: explicit points; mode 1: fused render.  Two shapes of the same kernel:
//   WAVES = 4 -- 64 points per 256-thread workgroup, TWO workgroups per CU (each at most 80 KiB of LDS: its own two-buffer weight ring, the
//     decoder's biases / heads, latent rows).  The two waves of a SIMD then belong to DIFFERENT workgroups: they share no barrier, drift
//     out of phase, and whatever one is doing at a chunk rendezvous or a layer start the other's MFMAs fill the matrix pipe.  Costs twice the
//     L2 -> LDS weight traffic per point (8 B/clk/CU).  Taken whenever it fits: the ray (render mode) inside 64 points, LDS <= 80 KiB.
//   WAVES = 8 -- 128 points per 512-thread workgroup, one per CU; both waves of a SIMD run the same program between the same barriers.
// The workgroup's latent rows are staged in LDS when its points belong to ONE object and the table has at most LDS_LAT_ROWS rows.
static int fwd16_waves_override() {
#ifdef SNR16_FORCE_WAVES
    return SNR16_FORCE_WAVES;
#else
    static const int v = [] { const char* e = getenv("SNR_FP32_WAVES"); return e ? atoi(e) : 0; }();      // diagnostic: 4 or 8
    return v;
#endif
}

template <int MODE, int WAVES, bool LATLDS, bool MASKS, bool DUMP>
static int launch16(const DecoderIO& io, const Layout& L, const Lds16& lo, const float* xyz, const float* viewdir, const RayGeom& g, float* rgb,
                    float* depth, float* acc, hipStream_t st) {
    auto kern = decoder_fwd16_kernel<MODE, WAVES, LATLDS, MASKS, DUMP>;
    static bool attr_set = false;          // (dynamic LDS beyond the default cap must be granted once per kernel)
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SNR_E_LAUNCH;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((io.n_points + WAVES * 16 - 1) / (WAVES * 16));
    kern<<<grid, WAVES * 64, (size_t)lo.total * 4, st>>>(io, L, lo, xyz, viewdir, g, rgb, depth, acc);
    return snr_check_launch_();
}

template <int WAVES>
static int launch16_w(int mode, const DecoderIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, float* rgb,
                      float* depth, float* acc, hipStream_t st) {
    const bool latlds = (io.points_per_obj % (WAVES * 16)) == 0 && L.n_lat <= LDS_LAT_ROWS;
    const bool masks = io.masks != nullptr;
    const Lds16 lo = make_lds16(WAVES, L.n_mfma_layers, latlds ? L.n_lat : LDS_LAT_ROWS + 1);
#define SNR_L16(M, LL, MK, DP) launch16<M, WAVES, LL, MK, DP>(io, L, lo, xyz, viewdir, g, rgb, depth, acc, st)
    if (mode == 0) {
        if (latlds) return masks ? SNR_L16(0, true, true, false) : SNR_L16(0, true, false, false);
        return masks ? SNR_L16(0, false, true, false) : SNR_L16(0, false, false, false);
    }
    if (latlds) return masks ? SNR_L16(1, true, true, false) : SNR_L16(1, true, false, false);
    return masks ? SNR_L16(1, false, true, false) : SNR_L16(1, false, false, false);
#undef SNR_L16
}

int snr_fp32_fwd16_launch_(int mode, const DecoderIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, float* rgb,
                           float* depth, float* acc, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (io.act) return SNR_E_UNSUPPORTED;          // (training dumps: snr_mlp.hip's kernel, whose dump stores are staged through LDS)
    const bool lat4 = (io.points_per_obj % 64) == 0 && L.n_lat <= LDS_LAT_ROWS;
    const Lds16 lo4 = make_lds16(4, L.n_mfma_layers, lat4 ? L.n_lat : LDS_LAT_ROWS + 1);
    bool four = lo4.total * 4 <= 80 * 1024 && (mode == 0 || (g.S <= 64 && 64 % g.S == 0));
    if (fwd16_waves_override() == 8) four = false;
    if (fwd16_waves_override() == 4 && !(mode == 0 || (g.S <= 64 && 64 % g.S == 0))) four = false;
    else if (fwd16_waves_override() == 4) four = true;
    return four ? launch16_w<4>(mode, io, L, xyz, viewdir, g, rgb, depth, acc, st) : launch16_w<8>(mode, io, L, xyz, viewdir, g, rgb, depth, acc, st);
}
