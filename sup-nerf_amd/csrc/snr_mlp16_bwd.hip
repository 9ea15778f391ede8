// Backward of the fused decoder / render kernels, exact fp32, TWO waves per SIMD: the counterpart of snr_mlp16.hip (forward) for
// snr_mlp_bwd.hip's one-wave-per-SIMD kernel.  Same mathematics, same transposed weight stream (snr_layout.h: L.bwd), same saved ReLU
// bits / sigma / rgb, same outputs (latent-gradient partials per workgroup, d xyz / d viewdir or d rays_o / d rays_d / d t).
//
//   * Workgroup = 4 waves = 64 consecutive sample points, 256 threads, 77 KiB of LDS -> two independent workgroups per CU; wave w owns
//     points 16 w .. 16 w + 15, lane (n = lane & 15, g = lane >> 4), accumulator register r of tile T = feature 16 T + 4 g + r of point n.
//   * Every layer is G_in = W^T G_out on v_mfma_f32_16x16x4_f32, k-outer over the previous layer's tiles like the forward: input tile T's
//     four operand registers are made from accP[T] -- the saved ReLU bit applied as v_bfe_i32 + v_and, two VALU instructions per value, no
//     accumulator read (the accumulators are VGPRs here) -- behind every second MFMA group of the tile before; the weight ring's LDS-DMA
//     pieces go out one per group with a scalar base; the last k-step deposits the finished sums in the dead previous set.
//   * fp32 MFMAs execute on the vector ALUs: non-matrix instructions are not hidden, so the boundary work is kept small --
//     the density head's term is one 64-fma pass before enc_shape^T, the latent-term gradient a DPP reduce-scatter over the wave's 16
//     points (128 VALU per latent layer).
//   * Latent-gradient partials: one row per 64-point workgroup ([tiles64][n_lat][256]; the 32x32x2 kernel writes one per 32 points): the four
//     waves' rows meet in LDS behind the next chunk rendezvous; summed by the same deterministic tree kernel.
// Not here: the training dumps (layer_grads) -- the exact-fp32 training backward stays on snr_mlp_bwd.hip's kernel.
#include "snr_mlp16_core.hpp"
#include "snr_host.hpp"

namespace snr {

constexpr int WBUFB = K_VIEW_PAD * KC;              // floats per ring buffer of the backward stream: 288 rows x 32 (36 KiB)
// LDS map of the backward (floats): [ring 0 | ring 1 | sigma_w 256 | part 64 | latent-gradient rows 4 x 256]; the composite scratch (start of the kernel) lies over ring
// buffer 1, the positional-encoding scratch (end of the kernel) over ring buffer 0
constexpr int LB_RING1 = WBUFB, LB_SIGW = 2 * WBUFB, LB_PART = LB_SIGW + 256, LB_RED = LB_PART + 64, LB_TOTAL = LB_RED + 4 * 256;
static_assert(LB_TOTAL * 4 <= 80 * 1024, "two workgroups per CU");
static_assert(4 * PE_WAVE16 <= WBUFB && 64 * COMP_STRIDE <= WBUFB, "scratch aliases");

// piece i (1 KiB) of this wave's slice of a backward chunk of `rows` rows: 8, 9 (288 rows) or 2 (64 rows) pieces per wave
__device__ __forceinline__ void piece_b(const Dma16& d, const float* __restrict__ g, const float* lds_dst, const float* lds_base, int rows, int i) {
    unsigned sl = (unsigned)(rows * 32) * (unsigned)d.wave;                 // rows * 128 B / 4 waves
    const int hi8 = i >> 3;                                                  // the ninth piece: past the immediate's range, moved into the base
    const char* sbase = reinterpret_cast<const char*>(g) + sl + hi8 * 8192;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(d.lds0 + (unsigned)((lds_dst - lds_base) * 4) + sl + 4096u + (unsigned)hi8 * 8192u);
    dma_piece(i & 7, d.voff, sbase, m0v);
}
__device__ __forceinline__ void chunk_b(const Dma16& d, const float* __restrict__ g, const float* lds_dst, const float* lds_base, int rows) {
#pragma unroll
    for (int i = 0; i < 9; ++i)
        if (i * 32 < rows) piece_b(d, g, lds_dst, lds_base, rows, i);
}

struct RingB { int cur; int aoff[2]; };
__device__ __forceinline__ void ring_turn_b(RingB& p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    p.cur ^= 1;
}

// accumulator value -> operand value with the saved ReLU bit applied: bit (8 (T & 3) + r) of the lane's pre-shifted word T >> 2
__device__ __forceinline__ float masked(float v, const uint32_t (&mw)[4], int T, int r) {
    const int keep = __builtin_amdgcn_sbfe((int)mw[T >> 2], 8 * (T & 3) + r, 1);
    return __uint_as_float(__float_as_uint(v) & (uint32_t)keep);
}

// One transposed layer.  NT output tiles (16: 256 rows; 18: enc_viewdir^T, 288 rows; 4: enc_xyz^T), NCH chunks of 32 k (8; 4 for rgb.0^T).
// FROM_ACC: the operand tiles are accP's, masked by mw (all ones: no activation); else the explicit tiles xin[2 NCH].
// On entry the layer's first chunk is in the current buffer; at its last chunk it requests `next_first` (next_rows rows; 0 = none).
// after_first_turn(): called behind the layer's first chunk rendezvous (the workgroup's latent-gradient rows of the layer before are complete then).
struct Nothing { __device__ __forceinline__ void operator()() const {} };
template <int NT, int NCH, bool FROM_ACC, class After = Nothing>
__device__ __forceinline__ void layer_b(f32x4 (&accP)[18], const f32x4* xin, RingB& ring, float* lds, const uint32_t (&mw)[4], const Dma16& dm,
                                        const float* base, const float* next_first, int next_rows, After&& after_first_turn = Nothing()) {
    f32x4 accC[18];
    f32x4 a0, a1;
    constexpr int rows = NT * 16, chunk_floats = rows * KC, NG = NT / 2;
    f32x4 xa, xb;
    if (FROM_ACC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xa[r] = masked(accP[0][r], mw, 0, r);
    } else xa = xin[0];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const float* src = (ch < NCH - 1) ? base + (ch + 1) * chunk_floats : next_first;
        const int nrows = (ch < NCH - 1) ? rows : next_rows;
        float* const dst = lds + (ring.cur ^ 1) * WBUFB;
        const float* wb = lds + ring.cur * WBUFB;
        first_pair(a0, a1, wb + ring.aoff[0]);
        auto first = [&](auto&& between) {        // (the first tile of the layer starts the sums: constant-0 C operand)
            if (ch == 0) tile_mma<NT, TM_ZERO>(accC, accP, xa, wb + ring.aoff[0], a0, a1, wb + ring.aoff[1], between);
            else tile_mma<NT, 0>(accC, accP, xa, wb + ring.aoff[0], a0, a1, wb + ring.aoff[1], between);
        };
        first([&](int gi) {
            if (gi * 32 < nrows) piece_b(dm, src, dst, lds, nrows, gi);
            if (gi == NG - 1) {
#pragma unroll
                for (int p = NG; p < 9; ++p) if (p * 32 < nrows) piece_b(dm, src, dst, lds, nrows, p);        // (more pieces than groups: the rest behind the last)
            }
            if (FROM_ACC) {
#pragma unroll
                for (int k = (gi * 4) / NG; k < ((gi + 1) * 4) / NG; ++k) xb[3 - k] = masked(accP[2 * ch + 1][3 - k], mw, 2 * ch + 1, 3 - k);
            } else if (gi == 0) xb = xin[2 * ch + 1];
        });
        if (ch == NCH - 1) {
            tile_mma<NT, TM_LAST>(accC, accP, xb, wb + ring.aoff[1], a0, a1, nullptr);
        } else {
            tile_mma<NT, 0>(accC, accP, xb, wb + ring.aoff[1], a0, a1, nullptr, [&](int gi) {
                if (FROM_ACC) {
#pragma unroll
                    for (int k = (gi * 4) / NG; k < ((gi + 1) * 4) / NG; ++k) xa[3 - k] = masked(accP[2 * ch + 2][3 - k], mw, 2 * ch + 2, 3 - k);
                } else if (gi == 0) xa = xin[2 * ch + 2];
            });
        }
        ring_turn_b(ring);
        if (ch == 0) after_first_turn();
    }
}

// enc_xyz^T (256 -> 64 features: 4 output tiles, 8 chunks of 64 rows): four chunks at a time (one 32 KiB piece of the stream = one ring
// buffer), so the layer has two ring turns instead of eight -- a 64-row chunk is 32 MFMAs per wave, less than the latency of its own DMA.
template <class After>
__device__ __forceinline__ void layer_xyz_b(f32x4 (&accP)[18], RingB& ring, float* lds, const uint32_t (&mw)[4], const Dma16& dm, const float* base,
                                            After&& after_first_turn) {
    f32x4 accC[4];
    f32x4 a0, a1, xa, xb;
    constexpr int CH = K_XYZ_PAD * KC;          // floats per chunk
#pragma unroll
    for (int r = 0; r < 4; ++r) xa[r] = masked(accP[0][r], mw, 0, r);
#pragma unroll
    for (int sc = 0; sc < 2; ++sc) {
        float* const dst = lds + (ring.cur ^ 1) * WBUFB;
        const float* wb = lds + ring.cur * WBUFB;
        first_pair(a0, a1, wb + ring.aoff[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = 4 * sc + c;
            const float* wc = wb + c * CH;
            const float* wn = (c < 3) ? wc + CH + ring.aoff[0] : nullptr;
            auto mask_b = [&](int gi) {
#pragma unroll
                for (int k = 2 * gi; k < 2 * gi + 2; ++k) xb[3 - k] = masked(accP[2 * ch + 1][3 - k], mw, 2 * ch + 1, 3 - k);
            };
            auto mask_a = [&](int gi) {
                if (ch < 7) {
#pragma unroll
                    for (int k = 2 * gi; k < 2 * gi + 2; ++k) xa[3 - k] = masked(accP[2 * ch + 2][3 - k], mw, 2 * ch + 2, 3 - k);
                }
            };
            auto even = [&](int gi) {
                if (sc == 0) piece_b(dm, base + 4 * CH, dst, lds, 256, 2 * c + gi);
                mask_b(gi);
            };
            if (ch == 0) tile_mma<4, TM_ZERO>(accC, accP, xa, wc + ring.aoff[0], a0, a1, wc + ring.aoff[1], even);
            else tile_mma<4, 0>(accC, accP, xa, wc + ring.aoff[0], a0, a1, wc + ring.aoff[1], even);
            if (ch == 7) tile_mma<4, TM_LAST>(accC, accP, xb, wc + ring.aoff[1], a0, a1, wn);
            else tile_mma<4, 0>(accC, accP, xb, wc + ring.aoff[1], a0, a1, wn, mask_a);
        }
        ring_turn_b(ring);
        if (sc == 0) after_first_turn();
    }
}

// Latent-term gradient of one layer: every feature of the 16 finished accumulator tiles summed over the wave's 16 points (the lanes of a DPP
// row), as a reduce-scatter on the VALU: step A pairs lanes across bit 3 of the row index (row_mirror) and tiles T | T + 8, step B across
// bit 2 (row_half_mirror) and T | T + 4 -- the two halves of a pair sum different register sets (bank_mask), so every step halves the live
// registers --, C and D are plain sums inside a quad.  Lane i of a row ends with the sums of tiles 8 b3 + 4 b2 + {0..3}, i.e. features
// 16 T + 4 g + r; one lane per quad stores them: 128 VALU instructions instead of 256 for four row sums per value.  asm volatile keeps the
// statements in order (every DPP read >= 2 instructions behind the write of its source).  The wave's 256 sums go to its row of the workgroup's
// LDS block; behind the next chunk rendezvous the four rows are added in wave order and ONE row per 64-point workgroup goes to the workspace
// (flush_latent_rows): a quarter of the partial rows of a store per wave, and one level less in the reduction tree.
#define SNR_DPP_SELF(R, CTRL, BANK) asm volatile("v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:" BANK : "+v"(R))
#define SNR_DPP_FROM(R0, R1, CTRL, BANK) asm volatile("v_add_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:" BANK : "+v"(R0) : "v"(R1))
__device__ __forceinline__ void reduce16_store(const f32x4 (&acc)[18], float* dst /* the wave's 256-float row in LDS */, int lane, bool tile_live) {
    const int i = lane & 15, g = lane >> 4;
    // step A writes fresh registers (the accumulators stay: they are the next layer's operands): the two halves of a row write complementary banks
    float v[8][4];
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0x3" : "=&v"(v[T][r]) : "v"(acc[T][r]));
            asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(v[T][r]) : "v"(acc[T + 8][r]));
        }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) { SNR_DPP_SELF(v[T][r], "row_half_mirror", "0x5"); SNR_DPP_FROM(v[T][r], v[T + 4][r], "row_half_mirror", "0xa"); }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) SNR_DPP_SELF(v[T][r], "quad_perm:[1,0,3,2]", "0xf");
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) SNR_DPP_SELF(v[T][r], "quad_perm:[2,3,0,1]", "0xf");
    if ((i & 3) == 0) {
        const int T0 = 8 * ((i >> 3) & 1) + 4 * ((i >> 2) & 1);
#pragma unroll
        for (int t = 0; t < 4; ++t)       // (a wave tile past the end of the launch holds copies of the last point: it contributes zeros)
            *reinterpret_cast<f32x4*>(dst + 16 * (T0 + t) + 4 * g) = tile_live ? f32x4{v[t][0], v[t][1], v[t][2], v[t][3]} : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ void flush_latent_rows(const float* red /* lds + LB_RED */, float* __restrict__ dst /* the workgroup's 256 floats */, int tid) {
    const float s = ((red[tid] + red[256 + tid]) + red[512 + tid]) + red[768 + tid];
    dst[tid] = s;
}

// Tail of the render-mode backward for 16 points per wave (all 256 threads; 64 consecutive sample points, S divides 64; the four lane groups
// of a point hold the same values): snr_device.hpp's ray_grad_tail for this lane layout.
__device__ __forceinline__ void ray_grad_tail16(const RayGeom& g, float* __restrict__ d_rays_o, float* __restrict__ d_rays_d, float* __restrict__ d_t,
                                                float* part, long long tile64, long long ray, long long obj, long long gp, bool live, float tval, float u,
                                                float zc, float gx, float gy, float gz, float hx, float hy, float hz, float gzc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, gg = lane >> 4;
    const bool box = g.z_mode == SNR_Z_BOX;
    const float sc_ = box ? g.xyz_mul : g.xyz_mul / g.xyz_div[obj];
    const float px = (g.m[0] * gx + g.m[3] * gy + g.m[6] * gz) * sc_;
    const float py = (g.m[1] * gx + g.m[4] * gy + g.m[7] * gz) * sc_;
    const float pz = (g.m[2] * gx + g.m[5] * gy + g.m[8] * gz) * sc_;
    const float qx = g.m[0] * hx + g.m[3] * hy + g.m[6] * hz;
    const float qy = g.m[1] * hx + g.m[4] * hy + g.m[7] * hz;
    const float qz = g.m[2] * hx + g.m[5] * hy + g.m[8] * hz;
    const float rdx = g.rays_d[ray * 3], rdy = g.rays_d[ray * 3 + 1], rdz = g.rays_d[ray * 3 + 2];
    float c[8] = {px, py, pz, tval * px + qx, tval * py + qy, tval * pz + qz, 0.f, 0.f};
    float dt = rdx * px + rdy * py + rdz * pz;
    if (g.flags & SNR_METRIC_Z) {
        const float zs = g.z_scale[obj];
        const float k = zc > 0.f ? gzc * zs * zs * tval / zc : 0.f;
        dt += k * (rdx * rdx + rdy * rdy + rdz * rdz);
        c[3] += k * tval * rdx; c[4] += k * tval * rdy; c[5] += k * tval * rdz;
    } else {
        dt += gzc;
    }
    if (box) { c[6] = dt * (1.f - u); c[7] = dt * u; }
    if (!(live && gg == 0)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = 0.f;
    }
    if (d_t && !box && live && gg == 0) d_t[gp] = dt;
    if (!(d_rays_o || d_rays_d)) return;
    const int S = g.S;
    const int G = S < 16 ? S : 16;      // lanes of group 0 of this wave that share a ray (S divides 64)
#pragma unroll
    for (int i = 0; i < 6; ++i) c[i] = group_sum(c[i], G);
    if (box) { c[6] = group_sum(c[6], G); c[7] = group_sum(c[7], G); }
    if (S <= 16) {
        if (live && gg == 0 && (n % S) == 0) ray_finish(g, ray, c, d_rays_o, d_rays_d);
        return;
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[wave * 8 + i] = c[i];
    }
    __syncthreads();
    const int waves_per_ray = S / 16;              // 2 or 4
    const int rays_here = 64 / S;
    if (tid < rays_here) {
        const long long rr = tile64 * rays_here + tid;
        if (rr < g.n_rays) {
            float s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s[i] = 0.f;
                for (int w = 0; w < waves_per_ray; ++w) s[i] += part[(tid * waves_per_ray + w) * 8 + i];
            }
            ray_finish(g, rr, s, d_rays_o, d_rays_d);
        }
    }
}

#ifdef SNR_STAMPS   /* diagnostic build: s_memtime at phase boundaries into the d_t buffer, 8 per 16-point wave tile (tools/_diag/stamps16.py) */
#define SNR16_BSTAMP(i) do { if (io.d_t && lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        reinterpret_cast<unsigned long long*>(io.d_t)[tile16 * 8 + (i)] = t_; } } while (0)
#else
#define SNR16_BSTAMP(i) do {} while (0)
#endif

#ifdef SNR16_NO_PRIO
#define SNR16_PRIO(p) do {} while (0)
#else
#define SNR16_PRIO(p) __builtin_amdgcn_s_setprio(p)       /* prologue and tail at high priority: see snr_mlp16.hip */
#endif

// MODE 0: explicit points (backward of SUPNeRF.forward).  MODE 1: fused render.
template <int MODE>
__global__ void __launch_bounds__(256, 2)
decoder_bwd16_kernel(BwdIO io, Layout L, const float* __restrict__ xyz, const float* __restrict__ viewdir, RayGeom gm) {
    __shared__ __attribute__((aligned(16))) float lds[LB_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const long long tile64 = blockIdx.x;
    const long long tile16 = tile64 * 4 + wave;
    const long long tile32 = tile64 * 2 + (wave >> 1);
    const long long gp_raw = tile64 * 64 + wave * 16 + n;
    const bool live = gp_raw < io.n_points;
    const long long gp = live ? gp_raw : io.n_points - 1;
    const int sb = io.sb, tb = io.tb;
    const int n_relu = n_relu_layers(sb, tb);
    const int li_encshape = sb + 1, li_view = sb + 2, li_last = sb + tb + 2;
    const bool tile_live = tile16 * 16 < io.n_points;
    const long long tile32m = (tile32 * 32 < io.n_points) ? tile32 : 0;       // (wave tiles past the end read tile 0's bits: results discarded)

    SNR16_BSTAMP(0);
    SNR16_PRIO(3);
    Dma16 dm;
    dm.voff = lane * 16u + 4096u;
    dm.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds);
    dm.wave = __builtin_amdgcn_readfirstlane(wave);
    RingB ring;
    ring.cur = 0;
    {
        const int sw = (n >> 1) & 7;
        ring.aoff[0] = n * KC + (((0 + g) ^ sw) << 2);
        ring.aoff[1] = n * KC + (((4 + g) ^ sw) << 2);
    }
    // the transposed stream (snr_layout.h): rgb.0^T 4 chunks of 256 rows, then per 256-wide layer in reverse 8 chunks (288 rows for
    // enc_viewdir^T), enc_xyz^T 8 chunks of 64 rows
    constexpr long long C256 = 256 * KC, C288 = K_VIEW_PAD * KC;
    const float* const stream = io.packed + L.bwd;
    auto layer_base = [&](int li) {       // li = li_last .. 1; 0 = enc_xyz^T
        const float* p = stream + 4 * C256;
        for (int l = li_last; l > li; --l) p += 8 * (l == li_view ? C288 : C256);
        return p;
    };
    auto rows_of = [&](int li) { return li == li_view ? K_VIEW_PAD : (li == 0 ? 4 * K_XYZ_PAD /* four chunks at a time: layer_xyz_b */ : 256); };

    // ---- start the stream; the density head's weights -> LDS (one row)
    chunk_b(dm, stream, lds, lds, 256);
    if (wave == 0) row_dma16(dm, io.packed + L.sigma_w, lds + LB_SIGW, lds);

    // ---- this lane's point and its upstream gradient
    float x, y, z, dx, dy, dz, tval = 0.f, zc = 0.f, uval = 0.f;
    long long ray = 0, obj = 0;
    if (MODE == 0) {
        x = xyz[gp * 3]; y = xyz[gp * 3 + 1]; z = xyz[gp * 3 + 2];
        dx = viewdir[gp * 3]; dy = viewdir[gp * 3 + 1]; dz = viewdir[gp * 3 + 2];
    } else {
        const PointId id = point_id(gm, tile64, 64, wave * 16 + n, live);
        ray = id.ray; obj = id.obj;
        const SamplePoint sp = make_sample(gm, id.ray, id.s, id.obj);
        x = sp.x; y = sp.y; z = sp.z; dx = sp.dx; dy = sp.dy; dz = sp.dz; zc = sp.zc; tval = sp.t; uval = sp.u;
    }
    float gs = 0.f, gr = 0.f, ggr = 0.f, gb = 0.f, gzc = 0.f;
    if (MODE == 0) {
        if (live) {
            gs = io.d_sigmas ? io.d_sigmas[gp] : 0.f;
            if (io.d_rgbs) { gr = io.d_rgbs[gp * 3]; ggr = io.d_rgbs[gp * 3 + 1]; gb = io.d_rgbs[gp * 3 + 2]; }
        }
    } else {
        float* comp = lds + LB_RING1;            // (over ring buffer 1: chunk 1 is requested behind the barriers below)
        if (g == 0) comp[(wave * 16 + n) * COMP_STRIDE + 5] = zc;
        __syncthreads();
        const int S = gm.S;
        const int rays_here = 64 / S;
        const bool white = gm.flags & SNR_WHITE_BKGD;
        for (int r = wave; r < rays_here; r += 4) {
            const long long rr = tile64 * rays_here + r;
            if (rr >= gm.n_rays) break;
            float* c0 = comp + r * S * COMP_STRIDE;
            const float* srow = io.sigmas + rr * S;
            const float* crow = io.rgbs + rr * S * 3;
            const float ur = io.d_rgb ? io.d_rgb[rr * 3] : 0.f, ug = io.d_rgb ? io.d_rgb[rr * 3 + 1] : 0.f,
                        ub = io.d_rgb ? io.d_rgb[rr * 3 + 2] : 0.f;
            const float ud = io.d_depth ? io.d_depth[rr] : 0.f, ua = io.d_acc ? io.d_acc[rr] : 0.f;
            composite_ray_bwd<1>(S, lane, white, ur, ug, ub, ud, ua,
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
            const float* c = comp + (wave * 16 + n) * COMP_STRIDE;
            gs = c[0]; gr = c[1]; ggr = c[2]; gb = c[3]; gzc = c[4];
        }
    }
    // softplus'(pre) = sigmoid(pre) = 1 - exp(-sigma)   (sigma = softplus(pre); exact 1 in fp32 past the threshold)
    const float dpre = gs * (1.f - expf(-io.sigmas[gp]));

    // the lane's ReLU bits of a layer, pre-shifted so that feature 16 T + 4 g + r is bit 8 (T & 3) + r of word T >> 2 (snr_mlp16.hip, store_masks16x4)
    auto load_bits = [&](int slot, uint32_t (&mw)[4]) {
        const uint4 m = io.masks[(tile32m * n_relu + slot) * 64 + 16 * (wave & 1) + n + 32 * (g & 1)];
        const int sh = 4 * (g >> 1);
        mw[0] = m.x >> sh; mw[1] = m.y >> sh; mw[2] = m.z >> sh; mw[3] = m.w >> sh;
    };

    SNR16_BSTAMP(1);
    // ---- colour head backward: g_h = W2^T d_rgb, masked by rgb.0's ReLU bits -> the eight operand tiles of rgb.0^T
    f32x4 xh[8];
    {
        uint32_t mw[4];
        load_bits(n_relu - 1, mw);
        const float* w2 = io.packed + L.rgb2_w;
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const f32x4 wr = *reinterpret_cast<const f32x4*>(w2 + 16 * T + 4 * g);
            const f32x4 wg = *reinterpret_cast<const f32x4*>(w2 + 128 + 16 * T + 4 * g);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w2 + 256 + 16 * T + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) xh[T][r] = masked(wr[r] * gr + wg[r] * ggr + wb[r] * gb, mw, T, r);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();             // chunk 0 and sigma_w have landed; every wave is done with the composite scratch (ring buffer 1)

    SNR16_PRIO(0);
    SNR16_BSTAMP(2);
    // ---- rgb.0^T : 128 -> 256
    f32x4 accP[18];
    uint32_t mw[4], mw_next[4];
    const uint32_t ones[4] = {~0u, ~0u, ~0u, ~0u};
    load_bits(relu_slot(li_last, sb), mw_next);          // the first boundary's bits (layer li_last's), requested before its chunks
    layer_b<16, 4, false>(accP, xh, ring, lds, ones, dm, stream, layer_base(li_last), rows_of(li_last));

#ifndef SNR16_TAILSTAMPS
    SNR16_BSTAMP(3);
#endif
    // ---- 256-wide layers in reverse: texture .., enc_viewdir, enc_shape, shape ..
    f32x4 gdir[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    int la_pending = -1;               // latent layer whose four wave rows wait in LDS for the next rendezvous
    auto flush = [&]() {
        if (la_pending >= 0) flush_latent_rows(lds + LB_RED, io.partial + (tile64 * L.n_lat + la_pending) * 256, tid);
        la_pending = -1;
    };
#pragma unroll 1
    for (int li = li_last; li >= 1; --li) {
        const bool relu = (li != li_encshape);
#pragma unroll
        for (int i = 0; i < 4; ++i) mw[i] = relu ? mw_next[i] : ~0u;
        if (li == li_encshape) {         // enc_shape's output also feeds the density head: + d_pre w_sigma (one pass)
            const float* ws = lds + LB_SIGW;
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + 16 * T + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) accP[T][r] = fmaf(dpre, wv[r], accP[T][r]);
            }
        }
        // the bits the NEXT boundary applies (layer li - 1's; enc_xyz's after the loop) are requested now, a whole layer ahead
        load_bits(li - 1 >= 1 ? relu_slot(li - 1, sb) : 0, mw_next);
        const float* base = layer_base(li);
        const float* nxt = (li - 1 >= 1) ? layer_base(li - 1) : layer_base(0);
        if (li == li_view) layer_b<18, 8, true>(accP, nullptr, ring, lds, mw, dm, base, nxt, rows_of(li - 1), flush);
        else layer_b<16, 8, true>(accP, nullptr, ring, lds, mw, dm, base, nxt, rows_of(li - 1), flush);
        // accP = gradient wrt the INPUT of layer li = previous output + latent term
        const int la = latent_after(li - 1, sb, tb);
        if (la >= 0 && io.partial) { reduce16_store(accP, lds + LB_RED + wave * 256, lane, tile_live); la_pending = la; }
        if (li == li_view) { gdir[0] = accP[16]; gdir[1] = accP[17];
#ifndef SNR16_TAILSTAMPS
            SNR16_BSTAMP(4);
#endif
        }
    }
    SNR16_BSTAMP(5);

    // ---- enc_xyz^T : 256 -> 64 positional-encoding features
    layer_xyz_b(accP, ring, lds, mw_next, dm, layer_base(0), flush);

    SNR16_BSTAMP(6);
    SNR16_PRIO(3);
    // ---- positional-encoding backward through the per-wave scratch rows (over ring buffer 0: the stream is done, every wave passed its last barrier)
    float* sc = lds + wave * PE_WAVE16 + n * PE_ROW;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[16 * T + 4 * g + r] = accP[T][r];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[64 + 16 * T + 4 * g + r] = gdir[T][r];
    __syncthreads();
#ifdef SNR16_TAILSTAMPS
    SNR16_BSTAMP(3);
#endif
    float gx = 0.f, gy = 0.f, gz = 0.f, hx = 0.f, hy = 0.f, hz = 0.f;
    auto pe_grad = [&](const float* row, int q, int n_freq, float sn, float cs, float& ax, float& ay, float& az) {
        const int a = q % 3, f = q / 3;
        const float v = ldexpf(row[3 + q] * cs - row[3 + 3 * n_freq + q] * sn, f);
        ax += a == 0 ? v : 0.f; ay += a == 1 ? v : 0.f; az += a == 2 ? v : 0.f;
    };
#pragma unroll 1
    for (int i = 0; i < 8; i += 2) {                   // two pairs per trip on packed arithmetic (pe_sincos2)
        const int q = 8 * g + i;
        if (q < 3 * XYZ_FREQ) {
            f32x2 sn, cs;
            pe_sincos2(f32x2{ldexpf(pick3(x, y, z, q % 3), q / 3), ldexpf(pick3(x, y, z, (q + 1) % 3), (q + 1) / 3)}, &sn, &cs);
            pe_grad(sc, q, XYZ_FREQ, sn[0], cs[0], gx, gy, gz);
            pe_grad(sc, q + 1, XYZ_FREQ, sn[1], cs[1], gx, gy, gz);
        }
    }
    {
        const int q = 3 * g;
        f32x2 sn, cs;
        pe_sincos2(f32x2{ldexpf(pick3(dx, dy, dz, q % 3), q / 3), ldexpf(pick3(dx, dy, dz, (q + 1) % 3), (q + 1) / 3)}, &sn, &cs);
        pe_grad(sc + 64, q, DIR_FREQ, sn[0], cs[0], hx, hy, hz);
        pe_grad(sc + 64, q + 1, DIR_FREQ, sn[1], cs[1], hx, hy, hz);
        float s1, c1;
        pe_sincos(ldexpf(pick3(dx, dy, dz, (q + 2) % 3), (q + 2) / 3), &s1, &c1);
        pe_grad(sc + 64, q + 2, DIR_FREQ, s1, c1, hx, hy, hz);
    }
    if (g == 0) { gx += sc[0]; gy += sc[1]; gz += sc[2]; hx += sc[64]; hy += sc[65]; hz += sc[66]; }
    gx += __shfl_xor(gx, 16, 64); gx += __shfl_xor(gx, 32, 64);
    gy += __shfl_xor(gy, 16, 64); gy += __shfl_xor(gy, 32, 64);
    gz += __shfl_xor(gz, 16, 64); gz += __shfl_xor(gz, 32, 64);
    hx += __shfl_xor(hx, 16, 64); hx += __shfl_xor(hx, 32, 64);
    hy += __shfl_xor(hy, 16, 64); hy += __shfl_xor(hy, 32, 64);
    hz += __shfl_xor(hz, 16, 64); hz += __shfl_xor(hz, 32, 64);
#ifdef SNR16_TAILSTAMPS
    SNR16_BSTAMP(4);
#endif

    if (MODE == 0) {
        if (live && g == 0) {
            if (io.d_xyz) { io.d_xyz[gp * 3] = gx; io.d_xyz[gp * 3 + 1] = gy; io.d_xyz[gp * 3 + 2] = gz; }
            if (io.d_dir) { io.d_dir[gp * 3] = hx; io.d_dir[gp * 3 + 1] = hy; io.d_dir[gp * 3 + 2] = hz; }
        }
        return;
    }
#ifdef SNR_STAMPS
    ray_grad_tail16(gm, io.d_rays_o, io.d_rays_d, nullptr, lds + LB_PART, tile64, ray, obj, gp, live, tval, uval, zc, gx, gy, gz, hx, hy, hz, gzc);
    SNR16_BSTAMP(7);
#else
    ray_grad_tail16(gm, io.d_rays_o, io.d_rays_d, io.d_t, lds + LB_PART, tile64, ray, obj, gp, live, tval, uval, zc, gx, gy, gz, hx, hy, hz, gzc);
#endif
}

}  // namespace snr

using namespace snr;

// supported: the fused render needs a ray inside 64 points; the latent gradient whole 64-point workgroups per object
int snr_fp32_bwd16_supported_(int mode, const BwdIO& io, const RayGeom& g) {
    if (io.gdump) return 0;
    if (mode == 1 && !(g.S <= 64 && 64 % g.S == 0)) return 0;
    if (io.partial && (io.points_per_obj % 64) != 0) return 0;
    return 1;
}
int snr_fp32_bwd16_launch_(int mode, const BwdIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, void* stream_) {
    const unsigned grid = (unsigned)((io.n_points + 63) / 64);
    if (mode == 0) decoder_bwd16_kernel<0><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, xyz, viewdir, g);
    else decoder_bwd16_kernel<1><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, nullptr, nullptr, g);
    return snr_check_launch_();
}
