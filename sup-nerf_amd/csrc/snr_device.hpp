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

// Two angles at once: the same operations per element (bit-identical results), the multiplies and fused multiply-adds as packed
// instructions (v_pk_mul_f32 / v_pk_fma_f32: two values per issue slot) -- the encodings are ~21 sin / cos pairs per lane in front of every
// tile's first MFMA, and with one wave per SIMD that phase is paid in full (tools/stamps.py: 8.6 k of a tile's 127 k cycles).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pe_sincos2(f32x2 a, f32x2* sn, f32x2* cs) {
    if (__builtin_expect(!(fabsf(a[0]) <= 8192.f && fabsf(a[1]) <= 8192.f), 0)) {
        float s0, c0, s1, c1;
        pe_sincos(a[0], &s0, &c0); pe_sincos(a[1], &s1, &c1);
        *sn = f32x2{s0, s1}; *cs = f32x2{c0, c1};
        return;
    }
    const f32x2 t = a * 0.636619772367581343f;
    const f32x2 k = {rintf(t[0]), rintf(t[1])};
    const f32x2 nk = -k;
    f32x2 r = __builtin_elementwise_fma(nk, f32x2{1.57079637050628662109375f, 1.57079637050628662109375f}, a);
    r = __builtin_elementwise_fma(nk, f32x2{-4.371138828673793e-8f, -4.371138828673793e-8f}, r);
    r = __builtin_elementwise_fma(nk, f32x2{-1.7763568394002505e-15f, -1.7763568394002505e-15f}, r);
    const f32x2 z = r * r;
    const f32x2 ps = __builtin_elementwise_fma(__builtin_elementwise_fma(f32x2{-1.9515295891e-4f, -1.9515295891e-4f}, z, f32x2{8.3321608736e-3f, 8.3321608736e-3f}), z,
                                                f32x2{-1.6666654611e-1f, -1.6666654611e-1f});
    const f32x2 s = __builtin_elementwise_fma(ps * z, r, r);
    const f32x2 pc = __builtin_elementwise_fma(__builtin_elementwise_fma(f32x2{2.443315711809948e-5f, 2.443315711809948e-5f}, z, f32x2{-1.388731625493765e-3f, -1.388731625493765e-3f}), z,
                                                f32x2{4.166664568298827e-2f, 4.166664568298827e-2f});
    const f32x2 c = __builtin_elementwise_fma(pc * z, z, __builtin_elementwise_fma(f32x2{-0.5f, -0.5f}, z, f32x2{1.0f, 1.0f}));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = (int)k[e];
        const float s1 = (q & 1) ? c[e] : s[e], c1 = (q & 1) ? s[e] : c[e];
        (*sn)[e] = (q & 2) ? -s1 : s1;
        (*cs)[e] = ((q + 1) & 2) ? -c1 : c1;
    }
}

// ------------------------------------------------------------------ sample points
// Geometry of one launch, decoded from snr_render_args (host fills it once per launch).
struct RayGeom {
    const float* rays_o;
    const float* rays_d;
    const float* t_vals;     // depths per z_mode; SNR_Z_BOX: the (N,S) jitter or null
    const float* xyz_div;
    const float* z_scale;
    float m[9];
    float xyz_mul;
    int z_mode;
    int flags;
    long long n_rays;
    long long rays_per_obj;
    int S;
    const float* box_half;   // SNR_Z_BOX: (B,3)
    unsigned long long rng_seed, rng_offset, rng_threads;
};

struct SamplePoint {
    float x, y, z;      // decoder-frame point
    float dx, dy, dz;   // decoder-frame unit direction
    float zc;           // depth used by the composite
    float t;            // raw depth along the ray
    float u;            // SNR_Z_BOX: position of the sample in the unit interval, t = near (1 - u) + far u
};

// ---- Philox4x32-10 (Salmon et al., SC'11; the constants every implementation shares).  One call gives the four words of one counter.
__device__ __forceinline__ void philox4x32_10(unsigned long long key, unsigned long long ctr_lo, unsigned long long ctr_hi, uint32_t (&out)[4]) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The jitter of sample point i when the caller passes none: what torch.rand_like of the (N,S) depth table holds at i for the device
// generator state (seed, offset) -- the reference's draw, src/renderer.py:40 -- so that a seeded run sees the same numbers with no (N,S)
// tensor and no extra launch.  torch's kernel: `threads` threads, thread j owns Philox subsequence j and takes one 4-word call per round
// of 4 * threads elements, word w of round k going to element k * 4 threads + w * threads + j; a word becomes (w + 1) 2^-32 (rocrand's
// uniform, in (0,1]) and 1.0 is folded to 0 (aten/src/ATen/native/cuda/DistributionTemplates.h).
__device__ __forceinline__ float box_jitter_rng(const RayGeom& g, long long i) {
    unsigned long long sub = (unsigned long long)i, round = 0;
    int word = 0;
    if (g.rng_threads) {
        if ((((unsigned long long)i | (g.rng_threads << 2)) >> 32) == 0) {       // every realistic launch: 32-bit divisions (a 64-bit one is ~150 instructions)
            const uint32_t thr = (uint32_t)g.rng_threads, per_round = thr << 2, i32 = (uint32_t)i;
            const uint32_t rd = i32 / per_round, r = i32 - rd * per_round, w = r / thr;
            round = rd; word = (int)w; sub = r - w * thr;
        } else {
            const unsigned long long per_round = g.rng_threads * 4ull;
            round = (unsigned long long)i / per_round;
            const unsigned long long r = (unsigned long long)i - round * per_round;
            word = (int)(r / g.rng_threads);
            sub = r - (unsigned long long)word * g.rng_threads;
        }
    }
    uint32_t w[4];
    philox4x32_10(g.rng_seed, g.rng_offset / 4ull + round, sub, w);
    const uint32_t v = word == 0 ? w[0] : (word == 1 ? w[1] : (word == 2 ? w[2] : w[3]));
    const float f = __fadd_rn(2.3283064e-10f, __fmul_rn((float)v, 2.3283064e-10f));
    return f == 1.0f ? 0.f : f;
}

// torch.minimum / torch.maximum: a NaN in either argument is the result
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fminf(a, b); }
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

// Family B's bounds (ray_box_intersection_tensor, src/utils.py:283-327, as prepare_sampled_rays calls it, src/renderer.py:95-108):
// slab test of the ray (o, d) against the box [-hb, hb]; rays that miss get near = far = -1.
struct BoxSlab {
    float inv[3], tmin[3], tmax[3];     // 1/d, (-hb - o)/d, (hb - o)/d per axis
    float t_near, t_far;
    bool hit;
};
__device__ __forceinline__ BoxSlab box_slab(const float (&o)[3], const float (&d)[3], const float (&hb)[3]) {
    BoxSlab b;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        b.inv[a] = __fdiv_rn(1.f, d[a]);
        b.tmin[a] = __fmul_rn(__fsub_rn(-hb[a], o[a]), b.inv[a]);
        b.tmax[a] = __fmul_rn(__fsub_rn(hb[a], o[a]), b.inv[a]);
    }
    const float l0 = nan_min(b.tmin[0], b.tmax[0]), l1 = nan_min(b.tmin[1], b.tmax[1]), l2 = nan_min(b.tmin[2], b.tmax[2]);
    const float h0 = nan_max(b.tmin[0], b.tmax[0]), h1 = nan_max(b.tmin[1], b.tmax[1]), h2 = nan_max(b.tmin[2], b.tmax[2]);
    b.t_near = nan_max(nan_max(l0, l1), l2);
    b.t_far = nan_min(nan_min(h0, h1), h2);
    b.hit = (b.t_far > b.t_near) && (b.t_far > 0.f);        // (t_far * hit) > 0, src/utils.py:316-317
    return b;
}

__device__ __forceinline__ void box_ray(const RayGeom& g, long long ray, long long obj, float (&o)[3], float (&d)[3], float (&hb)[3], float& zs) {
    zs = g.z_scale[obj];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a] = __fdiv_rn(g.rays_o[ray * 3 + a], zs);
        d[a] = g.rays_d[ray * 3 + a];
        hb[a] = g.box_half[obj * 3 + a];
    }
}

__device__ __forceinline__ void box_ray(const RayGeom& g, long long ray, float (&o)[3], float (&d)[3], float (&hb)[3], float& zs) {
    box_ray(g, ray, ray / g.rays_per_obj, o, d, hb, zs);
}

__device__ __forceinline__ float load_t(const RayGeom& g, long long ray, int s, long long obj) {
    long long idx = g.z_mode == SNR_Z_SHARED ? s : (g.z_mode == SNR_Z_PER_OBJECT ? obj * g.S + s : ray * g.S + s);
    return g.t_vals[idx];
}

// point s of ray `ray`:  p = M * (((o + d*t) / xyz_div) * xyz_mul),  dir = M * d
// (src/utils.py:165,472-495; src/renderer.py:111,441).  Composite depth is t, or the metric
// distance |p_sampling - o| * z_scale for SNR_METRIC_Z (src/renderer.py:114).
// SNR_Z_BOX: o is rays_o / z_scale, t comes from the ray's own box bounds and xyz_div is not applied (see the header).
// (`obj` = ray / rays_per_obj: the two-waves fp32 kernels derive it without a per-lane 64-bit division, ~150 VALU instructions)
__device__ __forceinline__ SamplePoint make_sample(const RayGeom& g, long long ray, int s, long long obj) {
    SamplePoint sp;
    float ox, oy, oz, t;
    const float dx = g.rays_d[ray * 3 + 0], dy = g.rays_d[ray * 3 + 1], dz = g.rays_d[ray * 3 + 2];
    const bool box = g.z_mode == SNR_Z_BOX;
    sp.u = 0.f;
    if (box) {
        float o[3], d[3], hb[3], zs;
        box_ray(g, ray, obj, o, d, hb, zs);
        const BoxSlab b = box_slab(o, d, hb);
        const float near = b.hit ? b.t_near : -1.f, far = b.hit ? b.t_far : -1.f;
        const float step = 1.f / (float)g.S;            // S is a power of two: linspace(0, 1 - 1/S, S)[s] = s / S exactly
        const float jit = g.t_vals ? g.t_vals[ray * g.S + s] : box_jitter_rng(g, ray * g.S + s);
        sp.u = __fadd_rn((float)s * step, __fmul_rn(jit, step));                              // src/renderer.py:37-40
        t = __fadd_rn(__fmul_rn(near, __fsub_rn(1.f, sp.u)), __fmul_rn(far, sp.u));           // :41
        ox = o[0]; oy = o[1]; oz = o[2];
    } else {
        ox = g.rays_o[ray * 3 + 0]; oy = g.rays_o[ray * 3 + 1]; oz = g.rays_o[ray * 3 + 2];
        t = load_t(g, ray, s, obj);
    }
    // o + d*t with separate multiply and add like the reference's broadcast ops (no fma contraction)
    float px = __fadd_rn(ox, __fmul_rn(dx, t));
    float py = __fadd_rn(oy, __fmul_rn(dy, t));
    float pz = __fadd_rn(oz, __fmul_rn(dz, t));
    sp.t = t;
    if (g.flags & SNR_METRIC_Z) {
        const float zs = g.z_scale[obj];
        float ex = __fmul_rn(__fsub_rn(px, ox), zs), ey = __fmul_rn(__fsub_rn(py, oy), zs), ez = __fmul_rn(__fsub_rn(pz, oz), zs);
        sp.zc = sqrtf(__fmaf_rn(ez, ez, __fmaf_rn(ey, ey, __fmul_rn(ex, ex))));      // torch.norm's fma chain (CPU), src/renderer.py:114
    } else {
        sp.zc = t;
    }
    if (!box) {
        const float dv = g.xyz_div[obj];
        px = __fdiv_rn(px, dv); py = __fdiv_rn(py, dv); pz = __fdiv_rn(pz, dv);
    }
    px = __fmul_rn(px, g.xyz_mul);
    py = __fmul_rn(py, g.xyz_mul);
    pz = __fmul_rn(pz, g.xyz_mul);
    sp.x = g.m[0] * px + g.m[1] * py + g.m[2] * pz;
    sp.y = g.m[3] * px + g.m[4] * py + g.m[5] * pz;
    sp.z = g.m[6] * px + g.m[7] * py + g.m[8] * pz;
    sp.dx = g.m[0] * dx + g.m[1] * dy + g.m[2] * dz;
    sp.dy = g.m[3] * dx + g.m[4] * dy + g.m[5] * dz;
    sp.dz = g.m[6] * dx + g.m[7] * dy + g.m[8] * dz;
    return sp;
}
__device__ __forceinline__ SamplePoint make_sample(const RayGeom& g, long long ray, int s) { return make_sample(g, ray, s, ray / g.rays_per_obj); }

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

// ------------------------------------------------------------------ sample point -> ray, backward
// One thread finishes one ray: c[0..2] = gradient wrt the sampling-frame origin, c[3..5] wrt the direction, c[6], c[7] = gradient wrt the
// box bounds near / far (SNR_Z_BOX, summed over the ray's samples).  SNR_Z_BOX: the bounds' gradient goes back through the slab test
// the way torch's autograd takes it through ray_box_intersection_tensor (src/utils.py:304-314: reciprocal, two products, minimum /
// maximum per axis, maximum / minimum across the axes; ties split evenly like torch.maximum's derivative), and the origin's through
// `rays_o / (obj_diag / 2)` (src/renderer.py:103).  A product of a zero gradient with an infinite 1/d (a direction component that is
// exactly 0) is taken as 0 here; torch makes a NaN of it.
__device__ __forceinline__ void ray_finish(const RayGeom& g, long long ray, float (&c)[8], float* __restrict__ d_rays_o, float* __restrict__ d_rays_d) {
    if (g.z_mode == SNR_Z_BOX) {
        float o[3], d[3], hb[3], zs;
        box_ray(g, ray, o, d, hb, zs);
        if (!(g.flags & SNR_BOX_DETACH)) {
            const BoxSlab b = box_slab(o, d, hb);
            if (b.hit) {
                float l[3], hi[3], gl[3], gh[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) { l[a] = fminf(b.tmin[a], b.tmax[a]); hi[a] = fmaxf(b.tmin[a], b.tmax[a]); }
                auto w_gt = [](float x, float y) { return x > y ? 1.f : (x == y ? 0.5f : 0.f); };      // share of x in max(x, y)
                const float m1 = fmaxf(l[0], l[1]), n1 = fminf(hi[0], hi[1]);
                const float wm1 = w_gt(m1, l[2]), wl0 = w_gt(l[0], l[1]);
                gl[0] = c[6] * wm1 * wl0; gl[1] = c[6] * wm1 * (1.f - wl0); gl[2] = c[6] * (1.f - wm1);
                const float wn1 = w_gt(hi[2], n1), wh0 = w_gt(hi[1], hi[0]);                             // share of x in min(x, y) = w_gt(y, x)
                gh[0] = c[7] * wn1 * wh0; gh[1] = c[7] * wn1 * (1.f - wh0); gh[2] = c[7] * (1.f - wn1);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float w_lo = w_gt(b.tmax[a], b.tmin[a]);        // share of tmin in min(tmin, tmax); its share in the max is 1 - that
                    const float g_tmin = gl[a] * w_lo + gh[a] * (1.f - w_lo), g_tmax = gl[a] * (1.f - w_lo) + gh[a] * w_lo;
                    float d_inv = 0.f;
                    if (g_tmin != 0.f) { c[a] -= g_tmin * b.inv[a]; d_inv += g_tmin * (-hb[a] - o[a]); }
                    if (g_tmax != 0.f) { c[a] -= g_tmax * b.inv[a]; d_inv += g_tmax * (hb[a] - o[a]); }
                    if (d_inv != 0.f) c[3 + a] -= d_inv * b.inv[a] * b.inv[a];
                }
            }
        }
        c[0] /= zs; c[1] /= zs; c[2] /= zs;
    }
    if (d_rays_o) { d_rays_o[ray * 3] = c[0]; d_rays_o[ray * 3 + 1] = c[1]; d_rays_o[ray * 3 + 2] = c[2]; }
    if (d_rays_d) { d_rays_d[ray * 3] = c[3]; d_rays_d[ray * 3 + 1] = c[4]; d_rays_d[ray * 3 + 2] = c[5]; }
}

// Tail of the render-mode backward kernels, called by all 256 threads of the workgroup (128 consecutive sample points, S divides 128; both
// half-waves of a wave hold the same 32 points): per lane the gradient wrt the decoder-frame point (gx, gy, gz) and direction
// (hx, hy, hz) and the composite's gradient wrt the depth it was given (gzc) -> d_t per sample, d_rays_o / d_rays_d per ray
// (p' = M (((o + t d) / div) mul), dir' = M d; segmented wave sums over the samples of a ray, LDS combine across waves, no atomics).
// `part`: 64 floats of LDS nobody else uses any more.
__device__ __forceinline__ void ray_grad_tail(const RayGeom& g, float* __restrict__ d_rays_o, float* __restrict__ d_rays_d, float* __restrict__ d_t,
                                              float* part, long long tile128, long long ray, long long gp, bool live, float tval, float u,
                                              float zc, float gx, float gy, float gz, float hx, float hy, float hz, float gzc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    const long long obj = ray / g.rays_per_obj;
    const bool box = g.z_mode == SNR_Z_BOX;
    const float sc_ = box ? g.xyz_mul : g.xyz_mul / g.xyz_div[obj];
    // M^T g
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
        // zc = | t d | zs  ->  d zc/d t = zs^2 t |d|^2 / zc ,  d zc/d d = zs^2 t^2 d / zc
        const float zs = g.z_scale[obj];
        const float k = zc > 0.f ? gzc * zs * zs * tval / zc : 0.f;
        dt += k * (rdx * rdx + rdy * rdy + rdz * rdz);
        c[3] += k * tval * rdx; c[4] += k * tval * rdy; c[5] += k * tval * rdz;
    } else {
        dt += gzc;
    }
    if (box) { c[6] = dt * (1.f - u); c[7] = dt * u; }          // t = near (1 - u) + far u
    if (!(live && h == 0)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = 0.f;
    }
#ifndef SNR_STAMPS      /* the diagnostic build borrows d_t as its stamp buffer */
    if (d_t && !box && live && h == 0) d_t[gp] = dt;
#endif
    if (!(d_rays_o || d_rays_d)) return;
    const int S = g.S;
    const int G = S < 32 ? S : 32;      // lanes of this wave that share a ray (S divides 128)
#pragma unroll
    for (int i = 0; i < 6; ++i) c[i] = group_sum(c[i], G);
    if (box) { c[6] = group_sum(c[6], G); c[7] = group_sum(c[7], G); }
    if (S <= 32) {
        if (live && h == 0 && (p % S) == 0) ray_finish(g, ray, c, d_rays_o, d_rays_d);
        return;
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[wave * 8 + i] = c[i];
    }
    __syncthreads();
    const int waves_per_ray = S / 32;              // 2 or 4
    const int rays_here = 128 / S;
    if (tid < rays_here) {
        const long long rr = tile128 * rays_here + tid;
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

}  // namespace snr
