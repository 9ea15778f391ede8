// Device-side building blocks shared by the gfx950 kernels: sample-point generation, positional
// encoding, and the wavefront alpha-composite (forward scan + analytic backward).
// Wavefront = 64 lanes throughout (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include "snr_layout.h"
#include "../../include/supnerf_hip.h"

namespace snr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float LAST_DELTA = 1e10f;   // src/utils.py:209
constexpr float TRANS_EPS = 1e-10f;   // src/utils.py:211

// Cross-row sums on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap), no LDS round trip like ds_bpermute:
//   sum_row_pairs : lane i <- v[i] + v[i ^ 16]   (rows of 16 lanes: 0+1, 2+3)
//   sum_halves    : lane i <- v[i] + v[i ^ 32]
// Through inline asm: __builtin_amdgcn_permlane{16,32}_swap returned the first output twice with this compiler.
__device__ __forceinline__ float sum_row_pairs(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float sum_halves(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// Sum over the 16 lanes of a DPP row (every lane gets the row total): four v_add_f32 with row_ror 8, 4, 2, 1 -- VALU only.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum(float v) {
    v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); v += dpp_mov<0x121>(v);
    return v;
}
// Sum over aligned groups of G = 2..32 lanes (G a power of two), total in every lane of the group.
__device__ __forceinline__ float group_sum(float v, int G) {
    if (G == 32) return sum_row_pairs(row_sum(v));
    if (G == 16) return row_sum(v);
    for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------ wave primitives
__device__ __forceinline__ float wave_sum(float v) {
    return sum_halves(sum_row_pairs(row_sum(v)));           // 6 VALU ops instead of 6 ds_bpermute round trips
}

// DPP moves whose lanes without a source (shifted in from outside the row / wave, or rows masked out) keep `old`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_or(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// lane k <- lane k-1 across the whole wave (wave_shr:1), lane 0 <- fill
__device__ __forceinline__ float wave_shift_up1(float v, float fill) { return dpp_or<0x138>(fill, v); }
__device__ __forceinline__ float wave_lane(float v, int uniform_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), uniform_lane));
}

// inclusive product scan over the 64 lanes: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then the row totals
// travel with row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3) -- six VALU ops, no ds_bpermute
__device__ __forceinline__ float wave_scan_mul(float v, int /*lane*/) {
    v *= dpp_or<0x111>(1.f, v); v *= dpp_or<0x112>(1.f, v); v *= dpp_or<0x114>(1.f, v); v *= dpp_or<0x118>(1.f, v);
    v *= dpp_or<0x142, 0xa>(1.f, v);
    v *= dpp_or<0x143, 0xc>(1.f, v);
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v) {
    v += dpp_or<0x111>(0.f, v); v += dpp_or<0x112>(0.f, v); v += dpp_or<0x114>(0.f, v); v += dpp_or<0x118>(0.f, v);
    v += dpp_or<0x142, 0xa>(0.f, v);
    v += dpp_or<0x143, 0xc>(0.f, v);
    return v;
}

// inclusive suffix sum over the 64 lanes (lane k gets sum_{j>=k} v_j): the prefix scan on the reversed wave
__device__ __forceinline__ float wave_suffix_sum(float v, int lane) {
    return __shfl(wave_scan_add(__shfl(v, 63 - lane, 64)), 63 - lane, 64);
}

// ------------------------------------------------------------------ positional-encoding sin / cos
// sin and cos of one angle for the encodings (|angle| <= 512 * |x| with x an object-normalised coordinate): Cody-Waite
// reduction by pi/2 in three fused steps, cephes minimax polynomials on [-pi/4, pi/4], quadrant fix-up -- ~25 VALU ops
// against ~100 of the library sincosf (whose large-argument path is never needed here).  Measured against float64 over
// |angle| <= 1024: max abs error 9.2e-8 (the correctly rounded fp32 result: 6.4e-8).  Beyond 8192 the library routine runs.
__device__ __forceinline__ void pe_sincos(float a, float* sn, float* cs) {
    if (__builtin_expect(!(fabsf(a) <= 8192.f), 0)) { sincosf(a, sn, cs); return; }
    const float k = rintf(a * 0.636619772367581343f);
    float r = fmaf(-k, 1.57079637050628662109375f, a);              // pi/2 = C1 + C2 + C3, fp32 pieces
    r = fmaf(-k, -4.371138828673793e-8f, r);
    r = fmaf(-k, -1.7763568394002505e-15f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float s = fmaf(ps * z, r, r);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// ------------------------------------------------------------------ sample points
// Geometry of one launch, decoded from snr_render_args (host fills it once per launch).
struct RayGeom {
    const float* rays_o;
    const float* rays_d;
    const float* t_vals;
    const float* xyz_div;
    const float* z_scale;
    float m[9];
    float xyz_mul;
    int z_mode;
    int flags;
    long long n_rays;
    long long rays_per_obj;
    int S;
};

struct SamplePoint {
    float x, y, z;      // decoder-frame point
    float dx, dy, dz;   // decoder-frame unit direction
    float zc;           // depth used by the composite
    float t;            // raw depth along the ray
};

__device__ __forceinline__ float load_t(const RayGeom& g, long long ray, int s) {
    long long obj = ray / g.rays_per_obj;
    long long idx = g.z_mode == SNR_Z_SHARED ? s : (g.z_mode == SNR_Z_PER_OBJECT ? obj * g.S + s : ray * g.S + s);
    return g.t_vals[idx];
}

// point s of ray `ray`:  p = M * (((o + d*t) / xyz_div) * xyz_mul),  dir = M * d
// (src/utils.py:165,472-495; src/renderer.py:111,441).  Composite depth is t, or the metric
// distance |p_sampling - o| * z_scale for SNR_METRIC_Z (src/renderer.py:114).
__device__ __forceinline__ SamplePoint make_sample(const RayGeom& g, long long ray, int s) {
    SamplePoint sp;
    long long obj = ray / g.rays_per_obj;
    const float ox = g.rays_o[ray * 3 + 0], oy = g.rays_o[ray * 3 + 1], oz = g.rays_o[ray * 3 + 2];
    const float dx = g.rays_d[ray * 3 + 0], dy = g.rays_d[ray * 3 + 1], dz = g.rays_d[ray * 3 + 2];
    const float t = load_t(g, ray, s);
    // o + d*t with separate multiply and add like the reference's broadcast ops (no fma contraction)
    float px = __fadd_rn(ox, __fmul_rn(dx, t));
    float py = __fadd_rn(oy, __fmul_rn(dy, t));
    float pz = __fadd_rn(oz, __fmul_rn(dz, t));
    sp.t = t;
    if (g.flags & SNR_METRIC_Z) {
        const float zs = g.z_scale[obj];
        float ex = __fmul_rn(__fsub_rn(px, ox), zs), ey = __fmul_rn(__fsub_rn(py, oy), zs), ez = __fmul_rn(__fsub_rn(pz, oz), zs);
        sp.zc = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)));
    } else {
        sp.zc = t;
    }
    const float dv = g.xyz_div[obj];
    px = __fmul_rn(__fdiv_rn(px, dv), g.xyz_mul);
    py = __fmul_rn(__fdiv_rn(py, dv), g.xyz_mul);
    pz = __fmul_rn(__fdiv_rn(pz, dv), g.xyz_mul);
    sp.x = g.m[0] * px + g.m[1] * py + g.m[2] * pz;
    sp.y = g.m[3] * px + g.m[4] * py + g.m[5] * pz;
    sp.z = g.m[6] * px + g.m[7] * py + g.m[8] * pz;
    sp.dx = g.m[0] * dx + g.m[1] * dy + g.m[2] * dz;
    sp.dy = g.m[3] * dx + g.m[4] * dy + g.m[5] * dz;
    sp.dz = g.m[6] * dx + g.m[7] * dy + g.m[8] * dz;
    return sp;
}

// ------------------------------------------------------------------ positional encoding
// Feature f of PE(v, L) (src/model_supnerf.py:155-161): f<3 -> v[f]; 3<=f<3+3L -> sin(2^i v[a]);
// then cos(2^i v[a]) with q = f-3 (or f-3-3L), i = q/3, a = q%3.  2^i * v is exact in fp32.
__device__ __forceinline__ float pick3(float a, float b, float c, int i) { return i == 0 ? a : (i == 1 ? b : c); }

// ------------------------------------------------------------------ alpha composite, forward
// One wave per ray; lane = sample within a 64-sample chunk.  `fetch(k, sig, cr, cg, cb, z, znext)`
// supplies sample k (znext only read for k < S-1).  Returns the five ray outputs in every lane.
struct RayOut { float r, g, b, depth, acc; };

template <class Fetch>
__device__ __forceinline__ RayOut composite_ray_fwd(int S, int lane, bool white, Fetch&& fetch) {
    float carry = 1.f;
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sw = 0.f, acc = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int k = base + lane;
        const bool valid = k < S;
        float sig = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, z = 0.f, zn = 0.f;
        if (valid) fetch(k, sig, cr, cg, cb, z, zn);
        const float delta = (k == S - 1) ? LAST_DELTA : zn - z;
        const float alpha = 1.f - expf(-fmaxf(sig, 0.f) * delta);
        const float T = valid ? (1.f - alpha) + TRANS_EPS : 1.f;
        const float incl = wave_scan_mul(T, lane);
        const float excl = wave_shift_up1(incl, 1.f);
        const float A = carry * excl;
        const float w = valid ? alpha * A : 0.f;
        sr += w * cr; sg += w * cg; sb += w * cb; sd += w * z; sw += w;
        if (k == S - 1) acc = A;
        carry *= wave_lane(incl, 63);
    }
    RayOut o;
    o.r = wave_sum(sr); o.g = wave_sum(sg); o.b = wave_sum(sb); o.depth = wave_sum(sd);
    const float wsum = wave_sum(sw);
    o.acc = wave_lane(acc, (S - 1) & 63);
    if (white) { const float bg = 1.f - wsum; o.r += bg; o.g += bg; o.b += bg; }
    return o;
}

// ------------------------------------------------------------------ alpha composite, backward
// Analytic gradient of the above for S <= 64*NCH.  `fetch` as before; `emit(k, d_sigma, d_cr, d_cg,
// d_cb, d_z)` receives the per-sample gradients (d_z includes both the depth and the delta paths).
template <int NCH, class Fetch, class Emit>
__device__ __forceinline__ void composite_ray_bwd(int S, int lane, bool white, float g_r, float g_g, float g_b,
                                                  float g_depth, float g_acc, Fetch&& fetch, Emit&& emit) {
    float sig[NCH], cr[NCH], cg[NCH], cb[NCH], z[NCH], delta[NCH], e[NCH], T[NCH], A[NCH], w[NCH], dw[NCH];
    float carry = 1.f, acc_last = 0.f;
    const float g_white = white ? (g_r + g_g + g_b) : 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k = c * 64 + lane;
        const bool valid = k < S;
        float zn = 0.f;
        sig[c] = cr[c] = cg[c] = cb[c] = z[c] = 0.f;
        if (valid) fetch(k, sig[c], cr[c], cg[c], cb[c], z[c], zn);
        delta[c] = (k == S - 1) ? LAST_DELTA : zn - z[c];
        e[c] = expf(-fmaxf(sig[c], 0.f) * delta[c]);
        const float alpha = 1.f - e[c];
        T[c] = valid ? (1.f - alpha) + TRANS_EPS : 1.f;
        const float incl = wave_scan_mul(T[c], lane);
        const float excl = wave_shift_up1(incl, 1.f);
        A[c] = carry * excl;
        w[c] = valid ? alpha * A[c] : 0.f;
        dw[c] = valid ? (g_r * cr[c] + g_g * cg[c] + g_b * cb[c] + g_depth * z[c] - g_white) : 0.f;
        if (k == S - 1) acc_last = A[c];
        carry *= wave_lane(incl, 63);
    }
    acc_last = wave_lane(acc_last, (S - 1) & 63);
    // reverse pass: R_k = sum_{j>k} dw_j w_j  (+ g_acc * A_{S-1} for k < S-1)
    float tail = 0.f;           // sum over later chunks
    float dd[NCH];
#pragma unroll
    for (int c = NCH - 1; c >= 0; --c) {
        const int k = c * 64 + lane;
        const bool valid = k < S;
        const float p = dw[c] * w[c];
        const float incl = wave_suffix_sum(p, lane);
        const float R = (incl - p) + tail + ((k < S - 1) ? g_acc * acc_last : 0.f);
        tail += wave_lane(incl, 0);
        const float dT = R / T[c];
        const float dalpha = dw[c] * A[c] - dT;
        const float s = fmaxf(sig[c], 0.f);
        const float dsig = (valid && sig[c] > 0.f) ? dalpha * delta[c] * e[c] : 0.f;
        dd[c] = (valid && k < S - 1) ? dalpha * s * e[c] : 0.f;
        // stash d_sigma in sig[c] (no longer needed)
        sig[c] = dsig;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k = c * 64 + lane;
        const bool valid = k < S;
        // d_z_k = w_k g_depth + d_delta_{k-1} - d_delta_k
        const float prev_chunk_last = (c > 0) ? wave_lane(dd[c > 0 ? c - 1 : 0], 63) : 0.f;
        const float prev = wave_shift_up1(dd[c], prev_chunk_last);
        const float dz = w[c] * g_depth + prev - dd[c];
        if (valid) emit(k, sig[c], w[c] * g_r, w[c] * g_g, w[c] * g_b, dz);
    }
}

}  // namespace snr
