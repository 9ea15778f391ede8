// Backward of the fused decoder / render kernels for gfx950.
//
// Same structure as the forward (snr_mlp.hip): one workgroup = 4 waves = 128 sample points, gradients
// stay in registers in the MFMA accumulator layout, and every layer is one transposed product
// G_in = W^T * G_out on v_mfma_f32_32x32x2_f32 with the transposed weight stream of the packed buffer.
// Nothing is recomputed: the forward saved one ReLU bit per hidden unit (208 B / point) plus sigma / rgb
// per point, which is all the chain rule needs when only the latent terms, the sample positions and the
// view directions are differentiated (optimisation mode; decoder weights are constants here).
//
//   composite backward (wave scan) -> colour head -> rgb.0^T -> texture^T .. -> enc_viewdir^T -> (+ density
//   head) -> enc_shape^T -> shape^T .. -> enc_xyz^T -> positional-encoding backward -> ray origin /
//   direction / depth gradients.  Gradients of the per-object latent terms are reduced over the 32 points
//   of a wave with a register reduce-scatter and written as per-tile partials (no atomics, deterministic);
//   a small second kernel sums the partials per object.
#include "snr_mlp_core.hpp"
#include "snr_host.hpp"

namespace snr {

#ifdef SNR_STAMPS   /* diagnostic build (tools/build_diag.sh): lane 0 of every live wave tile writes s_memtime at phase boundaries into the d_t buffer
                       (same slots as the split-bf16 backward: tools/stamps.py prints them) */
#define SNR32_BSTAMP(i) do { if (io.d_t && lane == 0 && tile_live) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        reinterpret_cast<unsigned long long*>(io.d_t)[tile32 * 16 + (i)] = t_; } } while (0)
#else
#define SNR32_BSTAMP(i) do {} while (0)
#endif

// accumulators -> operand registers with the saved ReLU bits applied (has_mask == false: pass through);
// `add` (nullable, wave-uniform) is a per-feature vector scaled by `scale` added first (density-head path).
// Between two layers the matrix pipe waits for this, so the common case is 3 VALU instructions per value: read, one v_bfe_i32 that
// turns bit k into 0 / ~0, one v_and.
template <int NT>
__device__ __forceinline__ void masked_to_operand(const f32x16 (&acc)[9], float (&in)[9][16], bool has_mask, const uint4& mask /* requested a layer ahead */,
                                                  const float* __restrict__ add, float scale, int h) {
    int m[4] = {-1, -1, -1, -1};
    if (has_mask) { m[0] = (int)mask.x; m[1] = (int)mask.y; m[2] = (int)mask.z; m[3] = (int)mask.w; }
    if (add) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(add + 32 * t + 8 * j + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[t][4 * j + e] + scale * av[e];
                    const int keep = __builtin_amdgcn_sbfe(m[t >> 1], (t & 1) * 16 + 4 * j + e, 1);
                    in[t][4 * j + e] = __uint_as_float(__float_as_uint(v) & (uint32_t)keep);
                }
            }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int keep = __builtin_amdgcn_sbfe(m[t >> 1], (t & 1) * 16 + r, 1);
                in[t][r] = __uint_as_float(__float_as_uint(acc[t][r]) & (uint32_t)keep);
            }
    }
}

template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[9]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// Sum the 8 x 16 accumulator registers of tiles 0..7 over the 32 points of the wave (lanes with equal
// lane>>5) with a butterfly reduce-scatter: 124 shuffles instead of 640.  Afterwards lane (p,h) holds the
// totals of features 8p + 4h .. +3, i.e. the wave holds all 256 features once: one coalesced 1 KiB store.
__device__ __forceinline__ void reduce_points_store(const f32x16 (&acc)[9], float* __restrict__ dst, int lane) {
    const int p = lane & 31, h = lane >> 5;
    float a64[64], a32[32], a16[16], a8[8], a4[4];
    {
        const bool up = p & 16;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float lo = acc[i >> 4][i & 15], hi = acc[(i + 64) >> 4][(i + 64) & 15];
            a64[i] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, 16, 64);
        }
    }
    {
        const bool up = p & 8;
#pragma unroll
        for (int i = 0; i < 32; ++i) a32[i] = (up ? a64[i + 32] : a64[i]) + __shfl_xor(up ? a64[i] : a64[i + 32], 8, 64);
    }
    {
        const bool up = p & 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) a16[i] = (up ? a32[i + 16] : a32[i]) + __shfl_xor(up ? a32[i] : a32[i + 16], 4, 64);
    }
    {
        const bool up = p & 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) a8[i] = (up ? a16[i + 8] : a16[i]) + __shfl_xor(up ? a16[i] : a16[i + 8], 2, 64);
    }
    {
        const bool up = p & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = (up ? a8[i + 4] : a8[i]) + __shfl_xor(up ? a8[i] : a8[i + 4], 1, 64);
    }
    f32x4 o = {a4[0], a4[1], a4[2], a4[3]};
    *reinterpret_cast<f32x4*>(dst + 8 * p + 4 * h) = o;
}

// MODE 0: explicit points (backward of SUPNeRF.forward).  MODE 1: fused render.
template <int MODE>
__global__ void __launch_bounds__(256, 1)
decoder_bwd_kernel(BwdIO io, Layout L, const float* __restrict__ xyz, const float* __restrict__ viewdir, RayGeom g) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    const long long tile128 = blockIdx.x;
    const long long tile32 = tile128 * 4 + wave;
    const long long gp_raw = tile128 * 128 + wave * 32 + p;
    const bool live = gp_raw < io.n_points;
    const long long gp = live ? gp_raw : io.n_points - 1;
    const int sb = io.sb, tb = io.tb;
    const int n_relu = n_relu_layers(sb, tb);
    const int li_encshape = sb + 1, li_view = sb + 2, li_last = sb + tb + 2;
    // wave tiles past the end of the launch (last workgroup): read tile 0's ReLU bits (results are discarded), store nothing
    const bool tile_live = tile32 * 32 < io.n_points;
    const long long tile32m = tile_live ? tile32 : 0;
    // training dumps of the points decoder go through the wave's positional-encoding scratch (idle until the encoding gradient at the
    // kernel's end): whole cache lines per store, see dump_operand_staged
    const int dump_rows = (int)((io.n_points - tile32m * 32) < 32 ? (io.n_points - tile32m * 32) : 32);
    float* const dump_scr = lds + LDS_SCRATCH + wave * PE_WAVE;

    // ---- start the transposed weight stream
    Pipe pipe;
    pipe_init(pipe, io.packed + L.bwd, lane);
    chunk_dma(pipe.next, lds, 256, tid);
    pipe.next += 256 * KC;

    SNR32_BSTAMP(0);
    // ---- this lane's point and its upstream gradient
    float x, y, z, dx, dy, dz, tval = 0.f, zc = 0.f, uval = 0.f;
    long long ray = 0;
    if (MODE == 0) {
        x = xyz[gp * 3]; y = xyz[gp * 3 + 1]; z = xyz[gp * 3 + 2];
        dx = viewdir[gp * 3]; dy = viewdir[gp * 3 + 1]; dz = viewdir[gp * 3 + 2];
    } else {
        ray = gp / g.S;
        const SamplePoint sp = make_sample(g, ray, (int)(gp - ray * g.S));
        x = sp.x; y = sp.y; z = sp.z; dx = sp.dx; dy = sp.dy; dz = sp.dz; zc = sp.zc; tval = sp.t; uval = sp.u;
    }
    float gs = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gzc = 0.f;
    if (MODE == 0) {
        if (live) {
            gs = io.d_sigmas ? io.d_sigmas[gp] : 0.f;
            if (io.d_rgbs) { gr = io.d_rgbs[gp * 3]; gg = io.d_rgbs[gp * 3 + 1]; gb = io.d_rgbs[gp * 3 + 2]; }
        }
    } else {
        float* comp = lds + LDS_COMP;
        if (lane < 32) comp[(wave * 32 + p) * COMP_STRIDE + 5] = zc;
        __syncthreads();
        const int S = g.S;
        const int rays_here = 128 / S;
        const bool white = g.flags & SNR_WHITE_BKGD;
        for (int r = wave; r < rays_here; r += 4) {
            const long long rr = tile128 * rays_here + r;
            if (rr >= g.n_rays) break;
            float* c0 = comp + r * S * COMP_STRIDE;
            const float* srow = io.sigmas + rr * S;
            const float* crow = io.rgbs + rr * S * 3;
            const float ur = io.d_rgb ? io.d_rgb[rr * 3] : 0.f, ug = io.d_rgb ? io.d_rgb[rr * 3 + 1] : 0.f,
                        ub = io.d_rgb ? io.d_rgb[rr * 3 + 2] : 0.f;
            const float ud = io.d_depth ? io.d_depth[rr] : 0.f, ua = io.d_acc ? io.d_acc[rr] : 0.f;
            if (S <= 64) composite_ray_bwd<1>(S, lane, white, ur, ug, ub, ud, ua,
                [&](int k, float& s_, float& r_, float& g_, float& b_, float& z_, float& zn_) {
                    s_ = srow[k]; r_ = crow[3 * k]; g_ = crow[3 * k + 1]; b_ = crow[3 * k + 2];
                    z_ = c0[k * COMP_STRIDE + 5];
                    zn_ = (k < S - 1) ? c0[(k + 1) * COMP_STRIDE + 5] : 0.f;
                },
                [&](int k, float ds, float dcr, float dcg, float dcb, float dzz) {
                    float* c = c0 + k * COMP_STRIDE;
                    c[0] = ds; c[1] = dcr; c[2] = dcg; c[3] = dcb; c[4] = dzz;
                });
            else composite_ray_bwd<2>(S, lane, white, ur, ug, ub, ud, ua,
                [&](int k, float& s_, float& r_, float& g_, float& b_, float& z_, float& zn_) {
                    s_ = srow[k]; r_ = crow[3 * k]; g_ = crow[3 * k + 1]; b_ = crow[3 * k + 2];
                    z_ = c0[k * COMP_STRIDE + 5];
                    zn_ = (k < S - 1) ? c0[(k + 1) * COMP_STRIDE + 5] : 0.f;
                },
                [&](int k, float ds, float dcr, float dcg, float dcb, float dzz) {
                    float* c = c0 + k * COMP_STRIDE;
                    c[0] = ds; c[1] = dcr; c[2] = dcg; c[3] = dcb; c[4] = dzz;
                });
        }
        __syncthreads();
        if (live) {
            const float* c = comp + (wave * 32 + p) * COMP_STRIDE;
            gs = c[0]; gr = c[1]; gg = c[2]; gb = c[3]; gzc = c[4];
        }
    }
    // softplus'(pre) = sigmoid(pre) = 1 - exp(-sigma)   (sigma = softplus(pre); exact 1 in fp32 past the threshold)
    const float dpre = gs * (1.f - expf(-io.sigmas[gp]));

    float in[9][16];
    f32x16 acc[9];
    float gdir[16];

    SNR32_BSTAMP(1);
    // ---- colour head backward: g_h = W2^T d_rgb, masked by rgb.0's ReLU
    {
        const uint4 mk = io.masks[(tile32m * n_relu + (n_relu - 1)) * 64 + lane];
        const uint32_t m[2] = {mk.x, mk.y};
        const float* w2 = io.packed + L.rgb2_w;
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
                    const float v = wr[e] * gr + wg[e] * gg + wb[e] * gb;
                    const bool on = (m[t >> 1] >> ((t & 1) * 16 + 4 * j + e)) & 1u;
                    in[t][4 * j + e] = on ? v : 0.f;
                }
            }
    }
    if constexpr (MODE == 0) { if (io.gdump && tile_live) dump_operand_staged<4>(in, io.gdump + ((long long)(li_last + 1) * io.n_points + tile32 * 32) * 256, dump_rows, dump_scr, lane); }
    else if (io.gdump && live) dump_operand<4>(in, io.gdump + ((long long)(li_last + 1) * io.n_points + gp) * 256, h);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    SNR32_BSTAMP(2);
    // ---- rgb.0^T : 128 -> 256   (the first boundary's ReLU bits, layer li_last's, are requested before its four chunks)
    uint4 mk_next = io.masks[(tile32m * n_relu + relu_slot(li_last, sb)) * 64 + lane];
    auto rows_of = [&](int li) { return li == li_view ? K_VIEW_PAD : (li == 0 ? K_XYZ_PAD : 256); };
    step<8, 9, true>(acc, in[0], pipe, lds, 256, tid);                 // (ZERO_C: the accumulators start from the constant 0)
    step<8, 9>(acc, in[1], pipe, lds, 256, tid);
    step<8, 9>(acc, in[2], pipe, lds, 256, tid);
    step<8, 9>(acc, in[3], pipe, lds, rows_of(li_last), tid);

    SNR32_BSTAMP(3);
    // ---- 256-wide layers in reverse: texture .., enc_viewdir, enc_shape, shape ..
#pragma unroll 1
    for (int li = li_last; li >= 1; --li) {
        const bool is_view = (li == li_view);
        const bool relu = (li != li_encshape);
        // acc = gradient wrt the OUTPUT of layer li; enc_shape's output also feeds the density head
        masked_to_operand<8>(acc, in, relu, mk_next, li == li_encshape ? io.packed + L.sigma_w : nullptr, dpre, h);
        // the ReLU bits the NEXT boundary applies (layer li-1's, enc_xyz's after the loop) are requested now, a whole layer ahead:
        // a load at the boundary itself would expose a memory round trip with the matrix pipe idle
        mk_next = io.masks[(tile32m * n_relu + (li - 1 >= 1 ? relu_slot(li - 1, sb) : 0)   /* (enc_shape: a valid slot, not applied) */) * 64 + lane];
        if constexpr (MODE == 0) { if (io.gdump && tile_live) dump_operand_staged<8>(in, io.gdump + ((long long)li * io.n_points + tile32 * 32) * 256, dump_rows, dump_scr, lane); }
    else if (io.gdump && live) dump_operand<8>(in, io.gdump + ((long long)li * io.n_points + gp) * 256, h);
        const int rows_after = rows_of(li - 1);
        if (is_view) {      // two instances of the layer body, each with compile-time chunk heights and tile count: a run-time "ninth tile?" /
                            // "how many DMA pieces?" inside every chunk costs scalar branches that nothing overlaps (one wave per SIMD)
            step<8, 9, true>(acc, in[0], pipe, lds, K_VIEW_PAD, tid, true);
            step<8, 9>(acc, in[1], pipe, lds, K_VIEW_PAD, tid, true);
            step<8, 9>(acc, in[2], pipe, lds, K_VIEW_PAD, tid, true);
            step<8, 9>(acc, in[3], pipe, lds, K_VIEW_PAD, tid, true);
            step<8, 9>(acc, in[4], pipe, lds, K_VIEW_PAD, tid, true);
            step<8, 9>(acc, in[5], pipe, lds, K_VIEW_PAD, tid, true);
            step<8, 9>(acc, in[6], pipe, lds, K_VIEW_PAD, tid, true);
            step<8, 9>(acc, in[7], pipe, lds, rows_after, tid, true);
        } else {
            step<8, 9, true>(acc, in[0], pipe, lds, 256, tid, false);
            step<8, 9>(acc, in[1], pipe, lds, 256, tid, false);
            step<8, 9>(acc, in[2], pipe, lds, 256, tid, false);
            step<8, 9>(acc, in[3], pipe, lds, 256, tid, false);
            step<8, 9>(acc, in[4], pipe, lds, 256, tid, false);
            step<8, 9>(acc, in[5], pipe, lds, 256, tid, false);
            step<8, 9>(acc, in[6], pipe, lds, 256, tid, false);
            step<8, 9>(acc, in[7], pipe, lds, rows_after, tid, false);
        }
        // acc = gradient wrt the INPUT of layer li = previous output + latent term
        const int la = latent_after(li - 1, sb, tb);
        if (la >= 0 && io.partial && tile_live) reduce_points_store(acc, io.partial + (tile32 * L.n_lat + la) * 256, lane);
        if (is_view) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gdir[r] = acc[8][r];
        }
        if (li_last - li < 6) SNR32_BSTAMP(4 + (li_last - li));
    }

    SNR32_BSTAMP(11);
    // ---- enc_xyz^T : 256 -> 64 positional-encoding features
    masked_to_operand<8>(acc, in, true, mk_next, nullptr, 0.f, h);
    if constexpr (MODE == 0) { if (io.gdump && tile_live) dump_operand_staged<8>(in, io.gdump + ((long long)0 * io.n_points + tile32 * 32) * 256, dump_rows, dump_scr, lane); }
    else if (io.gdump && live) dump_operand<8>(in, io.gdump + ((long long)0 * io.n_points + gp) * 256, h);
    step<2, 9, true>(acc, in[0], pipe, lds, 64, tid);
    step<2, 9>(acc, in[1], pipe, lds, 64, tid);
    step<2, 9>(acc, in[2], pipe, lds, 64, tid);
    step<2, 9>(acc, in[3], pipe, lds, 64, tid);
    step<2, 9>(acc, in[4], pipe, lds, 64, tid);
    step<2, 9>(acc, in[5], pipe, lds, 64, tid);
    step<2, 9>(acc, in[6], pipe, lds, 64, tid);
    step<2, 9>(acc, in[7], pipe, lds, 0, tid);

    SNR32_BSTAMP(12);
    // ---- positional-encoding backward through the per-wave scratch rows
    float* sc = lds + LDS_SCRATCH + wave * PE_WAVE + p * PE_ROW;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[32 * c + 8 * (r >> 2) + 4 * h + (r & 3)] = acc[c][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[64 + 8 * (r >> 2) + 4 * h + (r & 3)] = gdir[r];
    __syncthreads();
    float gx = 0.f, gy = 0.f, gz = 0.f, hx = 0.f, hy = 0.f, hz = 0.f;
#pragma unroll 1
    for (int i = 0; i < 15; ++i) {
        const int q = 15 * h + i, a = q % 3, f = q / 3;
        float sn, cs;
        pe_sincos(ldexpf(pick3(x, y, z, a), f), &sn, &cs);
        const float v = ldexpf(sc[3 + q] * cs - sc[3 + 3 * XYZ_FREQ + q] * sn, f);
        gx += a == 0 ? v : 0.f; gy += a == 1 ? v : 0.f; gz += a == 2 ? v : 0.f;
    }
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        const int q = 6 * h + i, a = q % 3, f = q / 3;
        float sn, cs;
        pe_sincos(ldexpf(pick3(dx, dy, dz, a), f), &sn, &cs);
        const float v = ldexpf(sc[64 + 3 + q] * cs - sc[64 + 3 + 3 * DIR_FREQ + q] * sn, f);
        hx += a == 0 ? v : 0.f; hy += a == 1 ? v : 0.f; hz += a == 2 ? v : 0.f;
    }
    if (h == 0) { gx += sc[0]; gy += sc[1]; gz += sc[2]; hx += sc[64]; hy += sc[65]; hz += sc[66]; }
    gx = sum_halves(gx); gy = sum_halves(gy); gz = sum_halves(gz);
    hx = sum_halves(hx); hy = sum_halves(hy); hz = sum_halves(hz);

    if (MODE == 0) {
        if (live && h == 0) {
            if (io.d_xyz) { io.d_xyz[gp * 3] = gx; io.d_xyz[gp * 3 + 1] = gy; io.d_xyz[gp * 3 + 2] = gz; }
            if (io.d_dir) { io.d_dir[gp * 3] = hx; io.d_dir[gp * 3 + 1] = hy; io.d_dir[gp * 3 + 2] = hz; }
        }
        return;
    }

    // ---- sample point -> ray: p' = M (((o + t d) / div) mul), dir' = M d
    SNR32_BSTAMP(13);
    ray_grad_tail(g, io.d_rays_o, io.d_rays_d, io.d_t, lds + LDS_COMP /* the composite scratch is free by now */, tile128, ray, gp, live, tval, uval, zc,
                  gx, gy, gz, hx, hy, hz, gzc);
}

}  // namespace snr

using namespace snr;

int snr_launch_reduce_latent_(const float* partial, float* scratch, long long tiles_per_obj, int n_lat, long long n_obj, float* d_latent,
                              void* stream);
long long snr_reduce_scratch_floats_(long long tiles_per_obj, int n_lat, long long n_obj);
int snr_bf16_supported_(int sb, int tb, long long points_per_obj);
int snr_bf16_launch_bwd_(int mode, const BwdIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, void* stream_);
// snr_mlp16_bwd.hip: the exact-fp32 backward with two waves per SIMD (16-point wave tiles)
int snr_fp32_bwd16_supported_(int mode, const BwdIO& io, const RayGeom& g);
int snr_fp32_bwd16_launch_(int mode, const BwdIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, void* stream_);

// workspace = partial latent gradients [tiles][n_lat][256] + the reduction tree's scratch; sized for the smallest tile any kernel
// writes a row for (32 points: a wave tile of the 32x32 kernels; snr_mlp16_bwd.hip writes one row per 64-point workgroup)
static size_t bwd_ws_bytes(int64_t n_points, int64_t points_per_obj, int sb, int tb) {
    const int64_t tiles = (n_points + 31) / 32;
    const int64_t ppo = points_per_obj > 0 ? points_per_obj : n_points;
    const int64_t n_obj = ppo > 0 ? (n_points + ppo - 1) / ppo : 1;
    const int64_t tree = snr_reduce_scratch_floats_((ppo + 31) / 32, sb + tb, n_obj);
    return (size_t)((tiles * (int64_t)(sb + tb) * 256 + tree) * sizeof(float) + 256);
}

// the exact-fp32 backward: the two-waves-per-SIMD kernel where it applies (no training dumps, a ray inside 64 points), else the
// one-wave 32x32x2 kernel of rounds 1-3 (-DSNR_BWD32: always, for A/B timing).  *tile = points per partial row.
static int launch_fp32_bwd(int mode, const BwdIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, void* stream_, int* tile) {
#ifndef SNR_BWD32
    if (snr_fp32_bwd16_supported_(mode, io, g)) { *tile = 64; return snr_fp32_bwd16_launch_(mode, io, L, xyz, viewdir, g, stream_); }
#endif
    *tile = 32;
    const unsigned grid = (unsigned)((io.n_points + 127) / 128);
    if (mode == 0) decoder_bwd_kernel<0><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, xyz, viewdir, g);
    else decoder_bwd_kernel<1><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, nullptr, nullptr, g);
    return snr_check_launch_();
}

extern "C" {

size_t snr_decoder_bwd_ws_bytes(int64_t n_points, int64_t points_per_obj, int sb, int tb) { return bwd_ws_bytes(n_points, points_per_obj, sb, tb); }

int snr_decoder_bwd(const float* xyz, const float* viewdir, const float* latent, const float* packed, const void* relu_masks,
                    const float* sigmas, const float* d_sigmas, const float* d_rgbs, int64_t n_points, int64_t points_per_obj, int sb,
                    int tb, float* d_latent, float* d_xyz, float* d_viewdir, float* layer_grads, void* workspace, size_t ws_bytes, int precision,
                    void* stream_) {
    if (n_points == 0) return SNR_OK;
    if (!xyz || !viewdir || !latent || !packed || !relu_masks || !sigmas) return SNR_E_ARG;
    if (sb < 0 || tb < 0 || sb > MAX_BLOCKS || tb > MAX_BLOCKS || n_points < 0) return SNR_E_ARG;
    if (points_per_obj < 1 || (n_points % points_per_obj) != 0) return SNR_E_SHAPE;
    const bool want_lat = d_latent && (sb + tb) > 0;
    if (want_lat) {
        if (points_per_obj % 32) return SNR_E_UNSUPPORTED;   // a wave tile must not straddle two objects
        if (!workspace || ws_bytes < bwd_ws_bytes(n_points, points_per_obj, sb, tb)) return SNR_E_WORKSPACE;
    }
    BwdIO io{};
    io.packed = packed; io.latent = latent; io.sb = sb; io.tb = tb; io.n_points = n_points; io.points_per_obj = points_per_obj;
    io.masks = (const uint4*)relu_masks; io.sigmas = sigmas; io.d_sigmas = d_sigmas; io.d_rgbs = d_rgbs;
    io.partial = want_lat ? (float*)workspace : nullptr;
    io.d_xyz = d_xyz; io.d_dir = d_viewdir;
    io.gdump = layer_grads;
    RayGeom g{};
    const Layout L = make_layout(sb, tb);
    int rc, tile = 32;
    if (precision == SNR_BF16X3) {
        if (!snr_bf16_supported_(sb, tb, points_per_obj)) return SNR_E_UNSUPPORTED;
        rc = snr_bf16_launch_bwd_(0, io, L, xyz, viewdir, g, stream_);
    } else if (precision == SNR_FP32) {
        rc = launch_fp32_bwd(0, io, L, xyz, viewdir, g, stream_, &tile);
    } else return SNR_E_ARG;
    if (rc != SNR_OK) return rc;
    if (want_lat)
        return snr_launch_reduce_latent_(io.partial, io.partial + ((n_points + tile - 1) / tile) * (int64_t)(sb + tb) * 256, points_per_obj / tile, sb + tb,
                                         n_points / points_per_obj, d_latent, stream_);
    return SNR_OK;
}

size_t snr_render_bwd_ws_bytes(const snr_render_args* a) {
    if (!a) return 0;
    return bwd_ws_bytes(a->n_rays * (int64_t)a->n_samples, a->rays_per_obj * (int64_t)a->n_samples, a->shape_blocks, a->texture_blocks);
}

int snr_render_bwd(const snr_render_args* a, const float* sigmas, const float* rgbs, const void* relu_masks, const float* d_rgb,
                   const float* d_depth, const float* d_acc, float* d_latent, float* d_rays_o, float* d_rays_d, float* d_t,
                   void* workspace, size_t ws_bytes, void* stream_) {
    RayGeom g;
    int rc = snr_fill_geom_(a, &g, 1);
    if (rc != SNR_OK) return rc;
    if (a->n_rays == 0) return SNR_OK;
    if (!sigmas || !rgbs || !relu_masks) return SNR_E_ARG;
    if (a->n_samples > 128 || (128 % a->n_samples) != 0) return SNR_E_UNSUPPORTED;
#ifndef SNR_STAMPS      /* the diagnostic build borrows d_t as its stamp buffer */
    if (d_t && a->z_mode != SNR_Z_PER_RAY) return SNR_E_UNSUPPORTED;
#endif
    const int sb = a->shape_blocks, tb = a->texture_blocks;
    const long long P = a->n_rays * a->n_samples;
    const long long ppo = a->rays_per_obj * a->n_samples;
    const bool want_lat = d_latent && (sb + tb) > 0;
    if (want_lat) {
        if (ppo % 32) return SNR_E_UNSUPPORTED;
        if (!workspace || ws_bytes < bwd_ws_bytes(P, ppo, sb, tb)) return SNR_E_WORKSPACE;
    }
    BwdIO io{};
    io.packed = a->packed; io.latent = a->latent; io.sb = sb; io.tb = tb; io.n_points = P; io.points_per_obj = ppo;
    io.masks = (const uint4*)relu_masks; io.sigmas = sigmas; io.rgbs = rgbs;
    io.d_rgb = d_rgb; io.d_depth = d_depth; io.d_acc = d_acc;
    io.partial = want_lat ? (float*)workspace : nullptr;
    io.d_rays_o = d_rays_o; io.d_rays_d = d_rays_d; io.d_t = d_t;
    const Layout L = make_layout(sb, tb);
    int tile = 32;
    if (a->precision == SNR_BF16X3) {
        if (!snr_bf16_supported_(sb, tb, ppo)) return SNR_E_UNSUPPORTED;
        rc = snr_bf16_launch_bwd_(1, io, L, nullptr, nullptr, g, stream_);
    } else if (a->precision == SNR_FP32) {
        rc = launch_fp32_bwd(1, io, L, nullptr, nullptr, g, stream_, &tile);
    } else return SNR_E_ARG;
    if (rc != SNR_OK) return rc;
    if (want_lat)
        return snr_launch_reduce_latent_(io.partial, io.partial + ((P + tile - 1) / tile) * (long long)(sb + tb) * 256, ppo / tile, sb + tb,
                                         a->n_rays / a->rays_per_obj, d_latent, stream_);
    return SNR_OK;
}

}  // extern "C"
