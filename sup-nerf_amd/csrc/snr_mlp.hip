// Fused decoder kernels for gfx950 (MI355X): positional encoding -> code-conditioned MLP on the
// fp32 MFMA pipe -> (optionally) wavefront alpha composite, in one launch.
//
// Design:
//   * One workgroup = 4 wavefronts = 128 consecutive sample points; one wave per SIMD so each wave
//     may use the whole 512-entry unified register file.
//   * The GEMMs are computed transposed, Y^T = W * X^T, with v_mfma_f32_32x32x2_f32: the weight slice
//     is the A operand (rows = output features), the 32 points of the wave are the B/C/D columns
//     (lane & 31 = point).  With that orientation the 32x32 accumulator tile of layer l *is* the B
//     operand of layer l+1 (register r of lane (p,h) holds feature 32t + 8(r>>2) + 4h + (r&3), exactly
//     the k a B operand needs when the A fragment is read with one 16-byte LDS load), so activations
//     never leave the registers: no LDS or HBM traffic between layers.
//   * Weights stream from L2 through a double-buffered LDS ring in 32-deep k-chunks (ROWS x 32 fp32,
//     XOR-swizzled 16-byte slots -> conflict-free ds_read_b128) filled by LDS-DMA, shared by the 4 waves.
//   * Bias / latent adds, ReLU, softplus, the 256->1 density head and the 128->3 colour head run on
//     the VALU in the accumulator layout; the composite is a 64-lane product scan.
#include "snr_mlp_core.hpp"
#include "snr_host.hpp"

namespace snr {

#ifdef SNR_STAMPS   /* diagnostic build (tools/build_diag.sh): thread 0 of every workgroup writes s_memtime at phase boundaries into the sigma buffer */
#define SNR32_STAMP(i) do { if (threadIdx.x == 0 && io.sigmas) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        reinterpret_cast<unsigned long long*>(io.sigmas)[(long long)blockIdx.x * 16 + 4 + (i)] = t_; } } while (0)
#else
#define SNR32_STAMP(i) do {} while (0)
#endif

// -------------------------------------------------------------------------------------------
// The decoder for the 32 points of this wave.  Inputs: point (x,y,z) and direction (dx,dy,dz) of
// lane's point p = lane & 31 (both half-waves hold the same point).  Outputs sigma, r, g, b valid in
// every lane.  All four waves of the workgroup must call it together (block-wide barriers inside).
// -------------------------------------------------------------------------------------------
template <bool STAGED, bool MASKS>      // MASKS: the launch saves the ReLU bits (io.masks).  STAGED (the points decoder, which is what trains): activation dumps through LDS, whole cache lines per store
__device__ __forceinline__ void decoder_forward_tile(const DecoderIO& io, const Layout& L, float* lds, long long gp /*clamped point id*/, bool live,
                                                     long long tile32, float x, float y, float z, float dx, float dy, float dz,
                                                     float& o_sigma, float& o_r, float& o_g, float& o_b) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int sb = io.sb, tb = io.tb;
    const int n_relu = n_relu_layers(sb, tb);
    // a workgroup covers 4 wave tiles; the last one of a launch may own tiles past the end (buffers are sized for ceil(P/32) tiles)
    const bool tile_live = tile32 * 32 < io.n_points;
    const int dump_rows = STAGED ? (int)((io.n_points - tile32 * 32) < 32 ? (io.n_points - tile32 * 32) : 32) : 0;       // rows of this wave's tile that exist
    float* const dump_scr = lds + LDS_SCRATCH + wave * PE_WAVE;      // the wave's positional-encoding scratch, idle after the prologue
    const float* bias = lds + LDS_BIAS;            // staged in the prologue
    const float* heads = bias + L.n_mfma_layers * 256;   // sigma_w (256) | sigma_b | rgb2_w (384) | rgb2_b, as in the packed stream
    const float* lat = io.latent + (gp / io.points_per_obj) * (long long)L.n_lat * 256;

    // one latent table for the whole workgroup (its 128 points are consecutive) and few enough rows: they are staged in LDS with the biases and
    // the epilogues read them from there (an LDS round trip at the layer boundary instead of a memory one)
    const long long wg_first = (long long)blockIdx.x * 128;
    const long long wg_last = wg_first + 127 < io.n_points ? wg_first + 127 : io.n_points - 1;
    const bool lat_in_lds = (wg_first / io.points_per_obj) == (wg_last / io.points_per_obj) && L.n_lat <= LDS_LAT_ROWS;
    const float* lat_wg = io.latent + (wg_first / io.points_per_obj) * (long long)L.n_lat * 256;
    float in[9][16];
    f32x16 acc[8];
    uint32_t mask[4];

    // ---- prologue: first weight chunk in flight while the positional encodings are computed
    Pipe pipe;
    pipe_init(pipe, io.packed + L.fwd, lane);
    {
        chunk_dma(pipe.next, lds, 256, tid);
        pipe.next += 256 * KC;
        // every layer's bias -> LDS (1 KiB per LDS-DMA instruction and wave, landed before the prologue's rendezvous): the accumulators
        // are initialised from there between layers, an LDS round trip with the matrix pipe idle instead of a memory one
        // (+ 3 KiB: sigma_w | sigma_b | rgb2_w | rgb2_b follow the biases in the packed stream, snr_layout.h)
        for (int r = wave; r < L.n_mfma_layers + 3; r += 4) {
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)(io.packed + L.bias + r * 256 + lane * 4), (lptr_t)(lds + LDS_BIAS + r * 256), 16, 0, 0);
        }
        if (lat_in_lds)
            for (int r = wave; r < L.n_lat; r += 4) {
                typedef const __attribute__((address_space(1))) void* gptr_t;
                typedef __attribute__((address_space(3))) void* lptr_t;
                __builtin_amdgcn_global_load_lds((gptr_t)(lat_wg + r * 256 + lane * 4), (lptr_t)(lds + LDS_LAT + r * 256), 16, 0, 0);
            }
        float* sc = lds + LDS_SCRATCH + wave * PE_WAVE + p * PE_ROW;
        // 30 (freq, axis) pairs of the xyz encoding, 15 per half-wave; 12 of the direction encoding, 6 each
#pragma unroll 1
        for (int i = 0; i < 15; ++i) {
            const int q = 15 * h + i;
            float sn, cs;
            pe_sincos(ldexpf(pick3(x, y, z, q % 3), q / 3), &sn, &cs);
            sc[3 + q] = sn; sc[3 + 3 * XYZ_FREQ + q] = cs;
        }
#pragma unroll 1
        for (int i = 0; i < 6; ++i) {
            const int q = 6 * h + i;
            float sn, cs;
            pe_sincos(ldexpf(pick3(dx, dy, dz, q % 3), q / 3), &sn, &cs);
            sc[64 + 3 + q] = sn; sc[64 + 3 + 3 * DIR_FREQ + q] = cs;
        }
        if (h == 0) {
            sc[0] = x; sc[1] = y; sc[2] = z; sc[63] = 0.f;
            sc[64] = dx; sc[65] = dy; sc[66] = dz;
#pragma unroll
            for (int f = D_DIR; f < 32; ++f) sc[64 + f] = 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // operand registers: register 4j+e of tile c <- feature 32c + 8j + 4h + e
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) in[c][r] = sc[32 * c + 8 * (r >> 2) + 4 * h + (r & 3)];
#pragma unroll
        for (int r = 0; r < 16; ++r) in[8][r] = sc[64 + 8 * (r >> 2) + 4 * h + (r & 3)];
    }

    SNR32_STAMP(0);      // prologue (encodings, staging) done
    // ---- enc_xyz: 64 -> 256
    acc_init_bias<8, 8>(acc, bias, h);
    step<8, 8>(acc, in[0], pipe, lds, 256, tid);
    step<8, 8>(acc, in[1], pipe, lds, 256, tid);
    {
        const int la = latent_after(0, sb, tb);
        if (lat_in_lds && la >= 0) epilogue<8, 8, MASKS>(acc, in, true, lds + LDS_LAT + la * 256, h, mask);
        else epilogue<8, 8, MASKS>(acc, in, true, la >= 0 ? lat + la * 256 : nullptr, h, mask);
        if (MASKS && tile_live) io.masks[(tile32 * n_relu + 0) * 64 + lane] = make_uint4(mask[0], mask[1], mask[2], mask[3]);
        if constexpr (STAGED) { if (io.act && tile_live) dump_operand_staged<8>(in, io.act + ((long long)0 * io.n_points + tile32 * 32) * 256, dump_rows, dump_scr, lane); }
        else if (io.act && live) dump_operand<8>(in, io.act + ((long long)0 * io.n_points + gp) * 256, h);
    }

    SNR32_STAMP(1);      // enc_xyz + its epilogue
    // ---- the 256-wide middle layers: shape blocks, enc_shape, enc_viewdir, texture blocks
    const int li_encshape = sb + 1, li_view = sb + 2, li_last = sb + tb + 2;
    o_sigma = 0.f;
#pragma unroll 1
    for (int li = 1; li <= li_last; ++li) {
        const bool is_view = (li == li_view);
        const int rows_after = (li == li_last) ? 128 : 256;
        acc_init_bias<8, 8>(acc, bias + li * 256, h);
        step<8, 8>(acc, in[0], pipe, lds, 256, tid);
        step<8, 8>(acc, in[1], pipe, lds, 256, tid);
        step<8, 8>(acc, in[2], pipe, lds, 256, tid);
        step<8, 8>(acc, in[3], pipe, lds, 256, tid);
        step<8, 8>(acc, in[4], pipe, lds, 256, tid);
        step<8, 8>(acc, in[5], pipe, lds, 256, tid);
        step<8, 8>(acc, in[6], pipe, lds, 256, tid);
        step<8, 8>(acc, in[7], pipe, lds, is_view ? 256 : rows_after, tid);
        if (is_view) step<8, 8>(acc, in[8], pipe, lds, rows_after, tid);
        const bool relu = (li != li_encshape);
        const int la = latent_after(li, sb, tb);
        if (lat_in_lds && la >= 0) epilogue<8, 8, MASKS>(acc, in, relu, lds + LDS_LAT + la * 256, h, mask);
        else epilogue<8, 8, MASKS>(acc, in, relu, la >= 0 ? lat + la * 256 : nullptr, h, mask);
        if (MASKS && relu && tile_live)
            io.masks[(tile32 * n_relu + relu_slot(li, sb)) * 64 + lane] = make_uint4(mask[0], mask[1], mask[2], mask[3]);
        if constexpr (STAGED) { if (io.act && tile_live) dump_operand_staged<8>(in, io.act + ((long long)li * io.n_points + tile32 * 32) * 256, dump_rows, dump_scr, lane); }
        else if (io.act && live) dump_operand<8>(in, io.act + ((long long)li * io.n_points + gp) * 256, h);
        if (li <= 6) SNR32_STAMP(1 + li);      // layer li + its epilogue (+ bias init of the next)
        if (li == li_encshape) {
            // density head: softplus(w_sigma . y + b)   (src/model_supnerf.py:257)
            const float* ws = heads;
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + 32 * t + 8 * j + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) part = fmaf(wv[e], in[t][4 * j + e], part);
                }
            const float pre = sum_halves(part) + heads[L.sigma_b - L.sigma_w];
            o_sigma = pre > 20.f ? pre : log1pf(expf(pre));
        }
    }

    SNR32_STAMP(8);
    // ---- rgb.0: 256 -> 128, ReLU;  rgb.2: 128 -> 3 on the VALU
    acc_init_bias<4, 8>(acc, bias + (li_last + 1) * 256, h);
    step<4, 8>(acc, in[0], pipe, lds, 128, tid);
    step<4, 8>(acc, in[1], pipe, lds, 128, tid);
    step<4, 8>(acc, in[2], pipe, lds, 128, tid);
    step<4, 8>(acc, in[3], pipe, lds, 128, tid);
    step<4, 8>(acc, in[4], pipe, lds, 128, tid);
    step<4, 8>(acc, in[5], pipe, lds, 128, tid);
    step<4, 8>(acc, in[6], pipe, lds, 128, tid);
    step<4, 8>(acc, in[7], pipe, lds, 0, tid);
    SNR32_STAMP(9);      // rgb.0's chunks
    epilogue<4, 8, MASKS>(acc, in, true, nullptr, h, mask);
    if (MASKS && tile_live) io.masks[(tile32 * n_relu + (n_relu - 1)) * 64 + lane] = make_uint4(mask[0], mask[1], 0u, 0u);
    if constexpr (STAGED) { if (io.act && tile_live) dump_operand_staged<4>(in, io.act + ((long long)(li_last + 1) * io.n_points + tile32 * 32) * 256, dump_rows, dump_scr, lane); }
    else if (io.act && live) dump_operand<4>(in, io.act + ((long long)(li_last + 1) * io.n_points + gp) * 256, h);
    {
        const float* w2 = heads + (L.rgb2_w - L.sigma_w);
        float pr = 0.f, pg = 0.f, pb = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n0 = 32 * t + 8 * j + 4 * h;
                const f32x4 wr = *reinterpret_cast<const f32x4*>(w2 + n0);
                const f32x4 wg = *reinterpret_cast<const f32x4*>(w2 + 128 + n0);
                const f32x4 wb = *reinterpret_cast<const f32x4*>(w2 + 256 + n0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = in[t][4 * j + e];
                    pr = fmaf(wr[e], v, pr); pg = fmaf(wg[e], v, pg); pb = fmaf(wb[e], v, pb);
                }
            }
        const float* b2 = heads + (L.rgb2_b - L.sigma_w);
        o_r = sum_halves(pr) + b2[0];
        o_g = sum_halves(pg) + b2[1];
        o_b = sum_halves(pb) + b2[2];
    }
}

// ===========================================================================================
// kernels
// ===========================================================================================
// MODE 0: explicit points (SUPNeRF.forward drop-in).  MODE 1: fused render (sampling + composite).
template <int MODE, bool MASKS>
__global__ void __launch_bounds__(256, 1)
decoder_fwd_kernel(DecoderIO io, Layout L, const float* __restrict__ xyz, const float* __restrict__ viewdir, RayGeom g,
                   float* __restrict__ out_rgb, float* __restrict__ out_depth, float* __restrict__ out_acc) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31;
    const long long tile128 = blockIdx.x;
    const long long gp_raw = tile128 * 128 + wave * 32 + p;
    const bool live = gp_raw < io.n_points;
    const long long gp = live ? gp_raw : io.n_points - 1;
    float x, y, z, dx, dy, dz, zc = 0.f;
    if (MODE == 0) {
        x = xyz[gp * 3]; y = xyz[gp * 3 + 1]; z = xyz[gp * 3 + 2];
        dx = viewdir[gp * 3]; dy = viewdir[gp * 3 + 1]; dz = viewdir[gp * 3 + 2];
    } else {
        const long long ray = gp / g.S;
        const SamplePoint sp = make_sample(g, ray, (int)(gp - ray * g.S));
        x = sp.x; y = sp.y; z = sp.z; dx = sp.dx; dy = sp.dy; dz = sp.dz; zc = sp.zc;
    }
    float sg, cr, cg, cb;
#ifdef SNR_STAMPS   /* diagnostic build (tools/build_diag.sh): the sigma buffer receives {s_memtime, s_memrealtime} at both ends of the workgroup */
    unsigned long long st_c0 = 0, st_r0 = 0;
    if (tid == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c0), "=s"(st_r0) :: "memory");
#endif
    decoder_forward_tile<MODE == 0, MASKS>(io, L, lds, gp, live, tile128 * 4 + wave, x, y, z, dx, dy, dz, sg, cr, cg, cb);
#ifdef SNR_STAMPS
    if (tid == 0 && io.sigmas) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        unsigned long long* o = reinterpret_cast<unsigned long long*>(io.sigmas) + tile128 * 16;     // [4 ..]: SNR32_STAMP phases
        o[0] = st_c0; o[1] = st_r0; o[2] = c1; o[3] = r1;
    }
#else
    if (live && lane < 32) {
        if (io.sigmas) io.sigmas[gp] = sg;
        if (io.rgbs) { io.rgbs[gp * 3] = cr; io.rgbs[gp * 3 + 1] = cg; io.rgbs[gp * 3 + 2] = cb; }
    }
#endif
    if (MODE == 1) {
        float* comp = lds + LDS_COMP;
        if (lane < 32) {
            float* c = comp + (wave * 32 + p) * COMP_STRIDE;
            c[0] = sg; c[1] = cr; c[2] = cg; c[3] = cb; c[4] = zc;
        }
        __syncthreads();
        const int S = g.S;
        const int rays_here = 128 / S;           // host guarantees 128 % S == 0
        const bool white = g.flags & SNR_WHITE_BKGD;
        for (int r = wave; r < rays_here; r += 4) {
            const long long ray = tile128 * rays_here + r;
            if (ray >= g.n_rays) break;
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
}

}  // namespace snr

using namespace snr;

int snr_bf16_supported_(int sb, int tb, long long points_per_obj);
int snr_bf16_launch_fwd_(int mode, const DecoderIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, float* rgb,
                         float* depth, float* acc, void* stream_);
// The exact-fp32 forward runs on the two-waves-per-SIMD kernel of snr_mlp16.hip (v_mfma_f32_16x16x4_f32, 16 points per wave: round 4).  This
// file's one-wave-per-SIMD kernel (v_mfma_f32_32x32x2_f32, rounds 1-3) still serves the TRAINING forward of the exact-fp32 step (activation
// dumps staged through LDS) and, with -DSNR_FWD32, every fp32 forward for A/B timing (tools/build_variant.sh).
int snr_fp32_fwd16_launch_(int mode, const DecoderIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, float* rgb,
                           float* depth, float* acc, void* stream_);

extern "C" {

int snr_precision_supported(int precision, int sb, int tb, int64_t points_per_obj) {
    if (sb < 0 || tb < 0 || sb > MAX_BLOCKS || tb > MAX_BLOCKS) return 0;
    if (precision == SNR_FP32) return 1;
    if (precision == SNR_BF16X3) return snr_bf16_supported_(sb, tb, points_per_obj);
    return 0;
}

int snr_decoder_fwd(const float* xyz, const float* viewdir, const float* latent, const float* packed, int64_t n_points,
                    int64_t points_per_obj, int sb, int tb, float* sigmas, float* rgbs, void* relu_masks, float* activations, int precision,
                    void* stream_) {
    if (!xyz || !viewdir || !latent || !packed) return SNR_E_ARG;
    if (activations && !relu_masks) return SNR_E_ARG;                         /* training dumps go with the ReLU bits */
    if (sb < 0 || tb < 0 || sb > MAX_BLOCKS || tb > MAX_BLOCKS || n_points < 0) return SNR_E_ARG;
    if (points_per_obj < 1 || (n_points % points_per_obj) != 0) return SNR_E_SHAPE;
    if (n_points == 0) return SNR_OK;
    DecoderIO io{packed, latent, sb, tb, (long long)n_points, (long long)points_per_obj, sigmas, rgbs, (uint4*)relu_masks, activations, false};
    RayGeom g{};
    const Layout L = make_layout(sb, tb);
    if (precision == SNR_BF16X3) {
        if (!snr_bf16_supported_(sb, tb, points_per_obj)) return SNR_E_UNSUPPORTED;
        return snr_bf16_launch_fwd_(0, io, L, xyz, viewdir, g, nullptr, nullptr, nullptr, stream_);
    }
    if (precision != SNR_FP32) return SNR_E_ARG;
#ifndef SNR_FWD32
    // (training dumps stay on this file's kernel: its LDS-staged dump stores write whole cache lines, 13.9 against 14.2 ms per fp32 step)
    if (!activations) return snr_fp32_fwd16_launch_(0, io, L, xyz, viewdir, g, nullptr, nullptr, nullptr, stream_);
#endif
    const unsigned grid = (unsigned)((n_points + 127) / 128);
    if (relu_masks) decoder_fwd_kernel<0, true><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, xyz, viewdir, g, nullptr, nullptr, nullptr);
    else decoder_fwd_kernel<0, false><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, xyz, viewdir, g, nullptr, nullptr, nullptr);
    return snr_check_launch_();
}

int snr_render_fwd(const snr_render_args* a, float* rgb, float* depth, float* acc_trans, float* sigmas, float* rgbs,
                   void* relu_masks, void* stream_) {
    RayGeom g;
    int rc = snr_fill_geom_(a, &g, 1);
    if (rc != SNR_OK) return rc;
    if (!rgb || !depth || !acc_trans) return SNR_E_ARG;
    if (a->n_samples > 128 || (128 % a->n_samples) != 0) return SNR_E_UNSUPPORTED;
    if (a->n_rays == 0) return SNR_OK;
    const long long P = a->n_rays * a->n_samples;
    DecoderIO io{a->packed, a->latent, a->shape_blocks, a->texture_blocks, P, a->rays_per_obj * a->n_samples, sigmas, rgbs,
                 (uint4*)relu_masks, nullptr, false};
    io.latent_bias = a->latent_bias;
    const Layout L = make_layout(a->shape_blocks, a->texture_blocks);
    if (a->precision == SNR_BF16X3) {
        if (!snr_bf16_supported_(a->shape_blocks, a->texture_blocks, a->rays_per_obj * a->n_samples)) return SNR_E_UNSUPPORTED;
        return snr_bf16_launch_fwd_(1, io, L, nullptr, nullptr, g, rgb, depth, acc_trans, stream_);
    }
    if (a->precision != SNR_FP32) return SNR_E_ARG;
#ifndef SNR_FWD32
    return snr_fp32_fwd16_launch_(1, io, L, nullptr, nullptr, g, rgb, depth, acc_trans, stream_);
#endif
    const unsigned grid = (unsigned)((P + 127) / 128);
    if (relu_masks) decoder_fwd_kernel<1, true><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, nullptr, nullptr, g, rgb, depth, acc_trans);
    else decoder_fwd_kernel<1, false><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, nullptr, nullptr, g, rgb, depth, acc_trans);
    return snr_check_launch_();
}

}  // extern "C"
