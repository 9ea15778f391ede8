// HBM-bound helper kernels for gfx950: weight packing, standalone alpha composite (forward and
// analytic backward), standalone sample encoding, and the small reductions used by the backward pass.
#include "snr_device.hpp"
#include "snr_host.hpp"

namespace snr {

// ============================================================================ weight packing
// W is nn.Linear layout (n_out, k_in).  Forward stream: chunk c holds rows n (n_out of them) x 32
// reduction columns k = 32c..32c+31 (zero beyond k_in).  Backward stream: chunk c holds rows k
// (rows_pad of them, zero beyond k_in) x 32 reduction columns n = 32c..32c+31.
__global__ void pack_fwd_kernel(const float* __restrict__ Wt, int n_out, int k_in, int n_chunks, float* __restrict__ dst) {
    const long long total = (long long)n_chunks * n_out * KC;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % KC);
        const int row = (int)((i / KC) % n_out);
        const int c = (int)(i / ((long long)KC * n_out));
        const int k = c * KC + kk;
        dst[(long long)c * n_out * KC + chunk_pos(row, kk)] = (k < k_in) ? Wt[(long long)row * k_in + k] : 0.f;
    }
}

__global__ void pack_bwd_kernel(const float* __restrict__ Wt, int n_out, int k_in, int rows_pad, float* __restrict__ dst) {
    const int n_chunks = n_out / KC;
    const long long total = (long long)n_chunks * rows_pad * KC;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % KC);
        const int row = (int)((i / KC) % rows_pad);
        const int c = (int)(i / ((long long)KC * rows_pad));
        const int n = c * KC + kk;
        dst[(long long)c * rows_pad * KC + chunk_pos(row, kk)] = (row < k_in) ? Wt[(long long)n * k_in + row] : 0.f;
    }
}

__global__ void copy_pad_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int n_pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) dst[i] = (i < n) ? src[i] : 0.f;
}

// ============================================================================ composite
__global__ void __launch_bounds__(256) composite_fwd_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                            const float* __restrict__ zv, int z_mode, int flags, long long n_rays,
                                                            long long rays_per_obj, int S, float* __restrict__ rgb,
                                                            float* __restrict__ depth, float* __restrict__ acc) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const bool white = flags & SNR_WHITE_BKGD;
    for (long long ray = wave0; ray < n_rays; ray += n_waves) {
        const float* zrow = zv + (z_mode == SNR_Z_SHARED ? 0 : (z_mode == SNR_Z_PER_OBJECT ? (ray / rays_per_obj) * S : ray * S));
        const float* srow = sigmas + ray * S;
        const float* crow = rgbs + ray * S * 3;
        RayOut o = composite_ray_fwd(S, lane, white, [&](int k, float& sg, float& cr, float& cg, float& cb, float& z, float& zn) {
            sg = srow[k]; cr = crow[3 * k]; cg = crow[3 * k + 1]; cb = crow[3 * k + 2];
            z = zrow[k]; zn = (k < S - 1) ? zrow[k + 1] : 0.f;
        });
        if (lane == 0) {
            rgb[ray * 3] = o.r; rgb[ray * 3 + 1] = o.g; rgb[ray * 3 + 2] = o.b;
            depth[ray] = o.depth; acc[ray] = o.acc;
        }
    }
}

// The same composite with 16 lanes per ray and 4 consecutive samples per lane (S <= 64, S a multiple of 4 -- every shipped config):
// every load is 16 bytes per lane (sigma, z: one each; colour: three), four rays per wave.  In-lane serial products over the lane's
// four samples, then an exclusive product scan over the 16 lanes of the DPP row (row_shr 1, 2, 4, 8), row sums for the outputs.
__global__ void __launch_bounds__(256) composite_fwd16_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                              const float* __restrict__ zv, int z_mode, int flags, long long n_rays,
                                                              long long rays_per_obj, int S, float* __restrict__ rgb,
                                                              float* __restrict__ depth, float* __restrict__ acc) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const bool white = flags & SNR_WHITE_BKGD;
    const int k0 = 4 * sub;
    const bool valid = k0 < S;
    for (long long base = wave0 * 4; base < n_rays; base += n_waves * 4) {
        const long long ray = base + grp;
        const bool live = ray < n_rays && valid;
        const long long rr = ray < n_rays ? ray : n_rays - 1;
        const float* zrow = zv + (z_mode == SNR_Z_SHARED ? 0 : (z_mode == SNR_Z_PER_OBJECT ? (rr / rays_per_obj) * S : rr * S));
        f32x4 sg = {0.f, 0.f, 0.f, 0.f}, z4 = sg, c0 = sg, c1 = sg, c2 = sg;
        if (live) {
            sg = *reinterpret_cast<const f32x4*>(sigmas + rr * S + k0);
            z4 = *reinterpret_cast<const f32x4*>(zrow + k0);
            const f32x4* cp = reinterpret_cast<const f32x4*>(rgbs + (rr * S + k0) * 3);
            c0 = cp[0]; c1 = cp[1]; c2 = cp[2];
        }
        const float zn = dpp_or<0x101>(0.f, z4[0]);              // row_shl:1 -- the next lane's first depth (unused where this lane ends the ray)
        const float zz[5] = {z4[0], z4[1], z4[2], z4[3], zn};
        const float cr[4] = {c0[0], c0[3], c1[2], c2[1]}, cg[4] = {c0[1], c1[0], c1[3], c2[2]}, cb[4] = {c0[2], c1[1], c2[0], c2[3]};
        float alpha[4], Al[4];
        float prod = 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            const float delta = (k == S - 1) ? LAST_DELTA : zz[e + 1] - zz[e];
            alpha[e] = 1.f - expf(-fmaxf(sg[e], 0.f) * delta);
            Al[e] = prod;
            prod *= live ? (1.f - alpha[e]) + TRANS_EPS : 1.f;
        }
        float incl = prod;
        incl *= dpp_or<0x111>(1.f, incl); incl *= dpp_or<0x112>(1.f, incl); incl *= dpp_or<0x114>(1.f, incl); incl *= dpp_or<0x118>(1.f, incl);
        const float excl = dpp_or<0x111>(1.f, incl);
        float sr = 0.f, sgc = 0.f, sb = 0.f, sd = 0.f, sw = 0.f, al = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float A = excl * Al[e];
            const float w = live ? alpha[e] * A : 0.f;
            sr += w * cr[e]; sgc += w * cg[e]; sb += w * cb[e]; sd += w * zz[e]; sw += w;
            if (live && k0 + e == S - 1) al = A;
        }
        sr = row_sum(sr); sgc = row_sum(sgc); sb = row_sum(sb); sd = row_sum(sd); sw = row_sum(sw); al = row_sum(al);
        if (white) { const float bg = 1.f - sw; sr += bg; sgc += bg; sb += bg; }
        if (sub == 0 && ray < n_rays) {
            rgb[ray * 3] = sr; rgb[ray * 3 + 1] = sgc; rgb[ray * 3 + 2] = sb;
            depth[ray] = sd; acc[ray] = al;
        }
    }
}

// ============================================================================ scene composite (multi-object pixels)
// One wave per pixel.  The pixel's n = Nb*S samples (Nb per-object lists) are merged by depth with a rank sort in LDS --
// every lane ranks its own elements against broadcast reads of the depth row -- then composited front to back with the same wave scan as every other composite here.  HBM-bound:
// 20 B per sample in, 20 B per pixel out; the n^2/64 compares per lane stay below the load time up to n ~ 512.
// `run` > 0: the pixel's n samples are n/run lists of `run` samples each, every list ascending in depth (one object's samples along its
// ray; a list of an object that does not cover the pixel is all -1).  Then a sample's rank is its position in its own list plus, for every
// other list, the number of smaller depths there -- two halving searches per list, O(n log(run) Nb) LDS reads per pixel instead of the n^2
// of the generic rank sort below.  The lists are CHECKED to be ascending; a pixel whose lists are not takes the generic path.
constexpr uint32_t SCENE_MARK = 0x7fc5ce4eu;    // "left to the general kernel": a quiet NaN with a payload

// Fast merge of a pass of up to 4 x 64 own elements into lists of RUN (a power of two) depths each: rank = # smaller elements over all lists.
// One branch-free halving search per (element, other list) with compile-time probe offsets (4 instructions a step); constant lists (the ray
// misses that object: -1 everywhere) by formula, ties included; an element's own list gives its place directly.  Returns false (wave-uniform)
// when an increasing list holds a depth equal to an element's -- the reference's equal-depth collapse then needs both bounds: general path.
template <int RUN>
__device__ __forceinline__ bool scene_merge_fast(const float* zs, int n, int n_runs, int base, int lane, const float (&zi)[4], const int (&rr)[4],
                                                 const int (&pp)[4], int (&lt)[4], int (&eb)[4], int (&ea)[4]) {
    bool ok = true;
    const int nc = (n - base + 63) >> 6;                                    // passes of 64 that hold elements (uniform)
    for (int q = 0; q < n_runs; ++q) {
        const float* zr = zs + q * RUN;
        const float z0 = zr[0], z1 = zr[RUN - 1];
        if (z0 == z1) {                                                     // uniform
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = zi[c];
                const bool eq = z0 == x, own = q == rr[c];
                lt[c] += (z0 < x) ? RUN : 0;
                eb[c] += eq ? (q < rr[c] ? RUN : (own ? pp[c] : 0)) : 0;
                ea[c] += eq ? (q > rr[c] ? RUN : (own ? RUN - 1 - pp[c] : 0)) : 0;
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c >= nc) continue;
            const float x = zi[c];
            const bool active = base + 64 * c + lane < n;
            if (RUN == 64 && q == (base >> 6) + c) {                        // uniform: this pass of 64 IS list q
                const float left = zr[(lane > 0 ? lane : 1) - 1], right = zr[lane < RUN - 1 ? lane + 1 : RUN - 2];
                ok = ok && (!active || ((lane == 0 || left < x) && (lane == RUN - 1 || right > x)));
                lt[c] += lane;
                continue;
            }
            const char* zb = reinterpret_cast<const char*>(zr);
            int off = 0;                                                    // bytes; elements before it are < x
#pragma unroll
            for (int step = RUN / 2; step > 0; step >>= 1) {
                const float m = *reinterpret_cast<const float*>(zb + off + (step - 1) * 4);
                off += (m < x) ? step * 4 : 0;
            }
            const float at = *reinterpret_cast<const float*>(zb + off);     // off <= (RUN - 1) * 4
            const int lbq = (off >> 2) + ((at < x) ? 1 : 0);
            const float nxt = *reinterpret_cast<const float*>(zb + (off + 4 < RUN * 4 ? off + 4 : off));
            const bool hit = (at == x) || (at < x && lbq < RUN && nxt == x);
            if (RUN == 64) {
                ok = ok && (!active || !hit);
                lt[c] += lbq;
            } else {                                                        // a pass may straddle lists: own list per lane
                const bool own = q == rr[c];
                const float left = zr[(pp[c] > 0 ? pp[c] : 1) - 1], right = zr[pp[c] < RUN - 1 ? pp[c] + 1 : RUN - 2];
                const bool own_ok = (pp[c] == 0 || left < x) && (pp[c] == RUN - 1 || right > x);
                ok = ok && (!active || (own ? own_ok : !hit));
                lt[c] += own ? pp[c] : lbq;
            }
        }
    }
    return __all(ok);
}

template <bool UPPER>
__device__ __forceinline__ int run_bound(const float* zr, int len, float v) {
    int lo = 0;
    while (len > 0) {
        const int half = len >> 1;
        const float m = zr[lo + half];
        const bool go = UPPER ? (m <= v) : (m < v);
        lo = go ? lo + half + 1 : lo;
        len = go ? len - half - 1 : half;
    }
    return lo;
}

// `run` > 0: the pixel's n samples are n/run lists of `run` samples each, every list ascending in depth (one object's samples along its
// ray; a list of an object that does not cover the pixel is all -1).  Then a sample's rank is its position in its own list plus, for every
// other list, the number of smaller depths there -- two binary searches per list, O(n log(run) Nb) LDS reads per pixel instead of the n^2
// of the generic rank sort below.  The lists are CHECKED to be ascending; a pixel whose lists are not takes the generic path.
template <bool MARKED_ONLY>
__global__ void __launch_bounds__(256) scene_general_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                              const float* __restrict__ zv, long long n_pixels, int n, int run, int flags,
                                                              float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) float scene_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* zs = scene_lds + (size_t)wave * 6 * n;          // unsorted depths | sorted sigma, r, g, b, z
    float* s_sig = zs + n; float* s_r = s_sig + n; float* s_g = s_r + n; float* s_b = s_g + n; float* s_z = s_b + n;
    const long long wave0 = (long long)blockIdx.x * 4 + wave;
    const long long n_waves = (long long)gridDim.x * 4;
    const bool white = flags & SNR_WHITE_BKGD;
    for (long long pix = wave0; pix < n_pixels; pix += n_waves) {
        if (MARKED_ONLY && __float_as_uint(rgb[pix * 3]) != SCENE_MARK) continue;     // the fast pass finished this pixel
        const float* zrow = zv + pix * n;
        const float* srow = sigmas + pix * n;
        const float* crow = rgbs + pix * n * 3;
        for (int i = lane; i < n; i += 64) zs[i] = zrow[i];
        __builtin_amdgcn_wave_barrier();
        // The reference scatters through searchsorted(sorted, z): samples with EQUAL depth land on one slot, the last one in
        // memory order wins and the group's other slots keep zero density / zero colour (torch scatter_ on the CPU).  Equal
        // depths are the rule for empty space (-1) and happen for ~1 % of pixels between real samples (fp32 depth grid), so
        // that is reproduced: the sorted depth row is complete, data goes to the group's first slot from its last member.
        bool merged = false;
        if (run > 1 && n % run == 0) {
            bool sorted = true;
            for (int i = lane; i < n; i += 64) sorted = sorted && ((i % run) == run - 1 || zs[i] <= zs[i + 1]);
            merged = __all(sorted);
        }
        if (merged) {
            const int n_runs = n / run;
            for (int base = 0; base < n; base += 256) {          // up to four own elements per pass; their payload loads fly under the searches
                float zi[4], sg[4], cr[4], cg[4], cb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = base + 64 * c + lane;
                    const bool on = i < n;
                    zi[c] = on ? zs[i] : 0.f;
                    sg[c] = on ? srow[i] : 0.f;
                    cr[c] = on ? crow[3 * i] : 0.f; cg[c] = on ? crow[3 * i + 1] : 0.f; cb[c] = on ? crow[3 * i + 2] : 0.f;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = base + 64 * c + lane;
                    if (i >= n) continue;
                    const int r = i / run, p = i - r * run;
                    int lt = 0, eb = 0, ea = 0;
                    for (int q = 0; q < n_runs; ++q) {
                        const float* zr = zs + q * run;
                        const int lb = run_bound<false>(zr, run, zi[c]);
                        const bool has_eq = lb < run && zr[lb] == zi[c];
                        const int ub = has_eq ? lb + 1 + run_bound<true>(zr + lb + 1, run - lb - 1, zi[c]) : lb;
                        lt += lb;
                        if (q < r) eb += ub - lb;
                        else if (q > r) ea += ub - lb;
                        else { eb += p - lb; ea += ub - p - 1; }
                    }
                    const int pos = lt + eb;
                    s_z[pos] = zi[c];
                    if (eb > 0) { s_sig[pos] = 0.f; s_r[pos] = 0.f; s_g[pos] = 0.f; s_b[pos] = 0.f; }
                    if (ea == 0) { s_sig[lt] = sg[c]; s_r[lt] = cr[c]; s_g[lt] = cg[c]; s_b[lt] = cb[c]; }
                }
            }
        }
        for (int base = 0; base < n && !merged; base += 256) {           // generic rank sort: up to four own elements per pass
            float zi[4]; int lt[4], eb[4], ea[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int i = base + 64 * c + lane; zi[c] = (i < n) ? zs[i] : 0.f; lt[c] = eb[c] = ea[c] = 0; }
            for (int j = 0; j < n; ++j) {
                const float zj = zs[j];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = base + 64 * c + lane;
                    lt[c] += (zj < zi[c]) ? 1 : 0;
                    eb[c] += (zj == zi[c] && j < i) ? 1 : 0;
                    ea[c] += (zj == zi[c] && j > i) ? 1 : 0;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = base + 64 * c + lane;
                if (i < n) {
                    const int pos = lt[c] + eb[c];
                    s_z[pos] = zi[c];
                    if (eb[c] > 0) { s_sig[pos] = 0.f; s_r[pos] = 0.f; s_g[pos] = 0.f; s_b[pos] = 0.f; }
                    if (ea[c] == 0) {
                        const int r = lt[c];
                        s_sig[r] = srow[i]; s_r[r] = crow[3 * i]; s_g[r] = crow[3 * i + 1]; s_b[r] = crow[3 * i + 2];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        RayOut o = composite_ray_fwd(n, lane, white, [&](int k, float& sg, float& cr, float& cg, float& cb, float& z, float& zn) {
            sg = s_sig[k]; cr = s_r[k]; cg = s_g[k]; cb = s_b[k];
            z = s_z[k]; zn = (k < n - 1) ? s_z[k + 1] : 0.f;
        });
        if (lane == 0) {
            rgb[pix * 3] = o.r; rgb[pix * 3 + 1] = o.g; rgb[pix * 3 + 2] = o.b;
            if (depth) depth[pix] = o.depth;
            if (acc) acc[pix] = o.acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- the fast pass -------------------------------------------------------------------------------------------------------------------
// Pixels whose n <= 256 samples are lists of 32 / 64 / 128 ascending depths without equal depths inside or across increasing lists (all
// but ~0.1 % of pixels): ranks by scene_merge_fast, scatter, composite.  Any other pixel is left to scene_general_kernel<true>, marked by a
// NaN with a payload in rgb[3 pix] (a pixel whose true result carries those bits is recomputed to the same value).  Keeping the general
// code out of this kernel keeps its registers low (6 waves per SIMD); the next pixel's 5 x 4 values per lane are requested one pixel ahead
// so that the memory latency is off the per-pixel chain LDS -> search -> scatter -> composite.
#ifndef SCENE_FAST_WAVES
#define SCENE_FAST_WAVES 6
#endif
template <int RUN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SCENE_FAST_WAVES, 8))) scene_fast_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ zv,
                                                         long long n_pixels, int n, int flags, float* __restrict__ rgb, float* __restrict__ depth,
                                                         float* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) float scene_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* zs = scene_lds + (size_t)wave * 6 * n;          // unsorted depths | sorted sigma, r, g, b, z
    float* s_sig = zs + n; float* s_r = s_sig + n; float* s_g = s_r + n; float* s_b = s_g + n; float* s_z = s_b + n;
    const long long wave0 = (long long)blockIdx.x * 4 + wave;
    const long long n_waves = (long long)gridDim.x * 4;
    const bool white = flags & SNR_WHITE_BKGD;
    const int n_runs = n / RUN;
    float zq[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f}, rq[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, bq[4] = {0.f, 0.f, 0.f, 0.f};
    auto request = [&](long long px) {                  // depths, densities and colours of pixel px, four per lane
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long i = px * n + 64 * c + lane;
            if (64 * c + lane < n) { zq[c] = zv[i]; sq[c] = sigmas[i]; rq[c] = rgbs[3 * i]; gq[c] = rgbs[3 * i + 1]; bq[c] = rgbs[3 * i + 2]; }
        }
    };
    if (wave0 < n_pixels) request(wave0);
    for (long long pix = wave0; pix < n_pixels; pix += n_waves) {
        float zi[4], sg[4], cr[4], cg[4], cb[4];
        int lt[4] = {0, 0, 0, 0}, eb[4] = {0, 0, 0, 0}, ea[4] = {0, 0, 0, 0}, rr[4], pp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 64 * c + lane;
            zi[c] = zq[c]; sg[c] = sq[c]; cr[c] = rq[c]; cg[c] = gq[c]; cb[c] = bq[c];
            if (i < n) zs[i] = zi[c];
            rr[c] = i / RUN; pp[c] = i - rr[c] * RUN;
        }
        if (pix + n_waves < n_pixels) request(pix + n_waves);
        __builtin_amdgcn_wave_barrier();
        bool sorted = true;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 64 * c + lane;
            if (i < n) sorted = sorted && (pp[c] == RUN - 1 || zi[c] <= zs[i + 1]);
        }
        bool ranked = __all(sorted);
        if (ranked) ranked = scene_merge_fast<RUN>(zs, n, n_runs, 0, lane, zi, rr, pp, lt, eb, ea);
        if (!ranked) {                                   // wave-uniform
            if (lane == 0) rgb[pix * 3] = __uint_as_float(SCENE_MARK);
            __builtin_amdgcn_wave_barrier();
            continue;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 64 * c + lane;
            if (i >= n) continue;
            const int pos = lt[c] + eb[c];
            s_z[pos] = zi[c];
            if (eb[c] > 0) { s_sig[pos] = 0.f; s_r[pos] = 0.f; s_g[pos] = 0.f; s_b[pos] = 0.f; }
            if (ea[c] == 0) { s_sig[lt[c]] = sg[c]; s_r[lt[c]] = cr[c]; s_g[lt[c]] = cg[c]; s_b[lt[c]] = cb[c]; }
        }
        __builtin_amdgcn_wave_barrier();
        RayOut o = composite_ray_fwd(n, lane, white, [&](int k, float& sg_, float& cr_, float& cg_, float& cb_, float& z, float& zn) {
            sg_ = s_sig[k]; cr_ = s_r[k]; cg_ = s_g[k]; cb_ = s_b[k];
            z = s_z[k]; zn = (k < n - 1) ? s_z[k + 1] : 0.f;
        });
        if (lane == 0) {
            rgb[pix * 3] = o.r; rgb[pix * 3 + 1] = o.g; rgb[pix * 3 + 2] = o.b;
            if (depth) depth[pix] = o.depth;
            if (acc) acc[pix] = o.acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NCH>
__global__ void __launch_bounds__(256) composite_bwd_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                            const float* __restrict__ zv, int z_mode, int flags, long long n_rays,
                                                            long long rays_per_obj, int S, const float* __restrict__ d_rgb,
                                                            const float* __restrict__ d_depth, const float* __restrict__ d_acc,
                                                            float* __restrict__ d_sigmas, float* __restrict__ d_rgbs,
                                                            float* __restrict__ d_z) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const bool white = flags & SNR_WHITE_BKGD;
    for (long long ray = wave0; ray < n_rays; ray += n_waves) {
        const float* zrow = zv + (z_mode == SNR_Z_SHARED ? 0 : (z_mode == SNR_Z_PER_OBJECT ? (ray / rays_per_obj) * S : ray * S));
        const float* srow = sigmas + ray * S;
        const float* crow = rgbs + ray * S * 3;
        const float gr = d_rgb ? d_rgb[ray * 3] : 0.f, gg = d_rgb ? d_rgb[ray * 3 + 1] : 0.f, gb = d_rgb ? d_rgb[ray * 3 + 2] : 0.f;
        const float gd = d_depth ? d_depth[ray] : 0.f, ga = d_acc ? d_acc[ray] : 0.f;
        composite_ray_bwd<NCH>(S, lane, white, gr, gg, gb, gd, ga,
            [&](int k, float& sg, float& cr, float& cg, float& cb, float& z, float& zn) {
                sg = srow[k]; cr = crow[3 * k]; cg = crow[3 * k + 1]; cb = crow[3 * k + 2];
                z = zrow[k]; zn = (k < S - 1) ? zrow[k + 1] : 0.f;
            },
            [&](int k, float ds, float dcr, float dcg, float dcb, float dz) {
                d_sigmas[ray * S + k] = ds;
                float* o = d_rgbs + (ray * S + k) * 3;
                o[0] = dcr; o[1] = dcg; o[2] = dcb;
                if (d_z) d_z[ray * S + k] = dz;
            });
    }
}

// ============================================================================ encode
// Block of 256 threads = 256 consecutive sample points.  Phase 1: one thread per point computes the
// sample (coalesced xyz / viewdir / z stores).  Phase 2: the block's 256x63 positional-encoding
// features are produced in flat output order so every store instruction is fully coalesced.
__global__ void __launch_bounds__(256) encode_kernel(RayGeom g, float* __restrict__ xyz, float* __restrict__ viewdir,
                                                     float* __restrict__ z_out, float* __restrict__ pe_xyz, uint8_t* __restrict__ hit) {
    __shared__ float sx[256][3];
    const long long P = g.n_rays * g.S;
    const long long base = blockIdx.x * 256ll;
    const long long gp = base + threadIdx.x;
    if (gp < P) {
        const long long ray = gp / g.S;
        const int s = (int)(gp - ray * g.S);
        SamplePoint sp = make_sample(g, ray, s);
        sx[threadIdx.x][0] = sp.x; sx[threadIdx.x][1] = sp.y; sx[threadIdx.x][2] = sp.z;
        if (xyz) { xyz[gp * 3] = sp.x; xyz[gp * 3 + 1] = sp.y; xyz[gp * 3 + 2] = sp.z; }
        if (viewdir) { viewdir[gp * 3] = sp.dx; viewdir[gp * 3 + 1] = sp.dy; viewdir[gp * 3 + 2] = sp.dz; }
        if (z_out) z_out[gp] = sp.zc;
        if (hit && s == 0) {        // the `intersect` map of prepare_sampled_rays (src/renderer.py:101-104)
            bool h = true;
            if (g.z_mode == SNR_Z_BOX) {
                float o[3], d[3], hb[3], zs;
                box_ray(g, ray, o, d, hb, zs);
                h = box_slab(o, d, hb).hit;
            }
            hit[ray] = h ? 1 : 0;
        }
    }
    if (!pe_xyz) return;
    __syncthreads();
    // Each (point, frequency, axis) angle is evaluated ONCE and gives both its sine and its cosine feature; the block's 256 x 63 features
    // are assembled in LDS and leave with 16-byte stores in flat output order (the block's slice of pe_xyz is contiguous and starts at a
    // multiple of 256 * 63 floats, i.e. 16-byte aligned whenever pe_xyz is).
    // (two passes of 128 points: 31.5 KiB of LDS per block keeps four blocks resident per CU)
    __shared__ __attribute__((aligned(16))) float pe[128 * D_XYZ];
    const long long n_blk = (P - base) < 256 ? (P - base) : 256;
    for (int half = 0; half < 2; ++half) {
        const int p0 = 128 * half;
        const int n_here = (int)(n_blk - p0 < 128 ? n_blk - p0 : 128);
        if (n_here <= 0) break;
        const int n_pairs = n_here * 3 * XYZ_FREQ;
        for (int i = threadIdx.x; i < n_pairs; i += 256) {
            const int p = i / (3 * XYZ_FREQ), q = i - p * (3 * XYZ_FREQ);
            float sn, cs;
            pe_sincos(ldexpf(sx[p0 + p][q % 3], q / 3), &sn, &cs);
            pe[p * D_XYZ + 3 + q] = sn;
            pe[p * D_XYZ + 3 + 3 * XYZ_FREQ + q] = cs;
        }
        if (threadIdx.x < n_here) {
#pragma unroll
            for (int a = 0; a < 3; ++a) pe[threadIdx.x * D_XYZ + a] = sx[p0 + threadIdx.x][a];
        }
        __syncthreads();
        const int total = n_here * D_XYZ;
        float* dst = pe_xyz + (base + p0) * D_XYZ;          // (base + p0) * 63 floats: a multiple of 128 * 63 * 4 bytes = 16-byte aligned with pe_xyz
        if ((((uintptr_t)dst) & 15) == 0) {
            const int n4 = total >> 2;
            for (int i = threadIdx.x; i < n4; i += 256) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(pe)[i];
            for (int i = (n4 << 2) + threadIdx.x; i < total; i += 256) dst[i] = pe[i];
        } else {
            for (int i = threadIdx.x; i < total; i += 256) dst[i] = pe[i];
        }
        __syncthreads();
    }
}

// Positional encodings of explicit points, as the weight-gradient products of the training step read them: out [P][96] = PE(xyz) (63 columns
// + 1 zero) | PE(dir) (27 + 5 zeros).  128 points per block; every angle is evaluated once (sine and cosine), the block's rows are assembled
// in LDS and leave with 16-byte stores.
__global__ void __launch_bounds__(256) pe_points_kernel(const float* __restrict__ xyz, const float* __restrict__ viewdir, long long P,
                                                        float* __restrict__ out) {
    constexpr int W = 96, NP = 128;
    __shared__ __attribute__((aligned(16))) float row[NP * W];
    __shared__ float pt[NP][6];
    const long long base = blockIdx.x * (long long)NP;
    const int n_here = (int)((P - base) < NP ? (P - base) : NP);
    for (int i = threadIdx.x; i < n_here * 6; i += 256) {
        const int p = i / 6, a = i - 6 * p;
        pt[p][a] = a < 3 ? xyz[(base + p) * 3 + a] : viewdir[(base + p) * 3 + a - 3];
    }
    __syncthreads();
    constexpr int NQ = 3 * XYZ_FREQ + 3 * DIR_FREQ;        // 30 + 12 (frequency, axis) pairs per point
    for (int i = threadIdx.x; i < n_here * NQ; i += 256) {
        const int p = i / NQ, q = i - p * NQ;
        float sn, cs;
        if (q < 3 * XYZ_FREQ) {
            pe_sincos(ldexpf(pt[p][q % 3], q / 3), &sn, &cs);
            row[p * W + 3 + q] = sn; row[p * W + 3 + 3 * XYZ_FREQ + q] = cs;
        } else {
            const int qd = q - 3 * XYZ_FREQ;
            pe_sincos(ldexpf(pt[p][3 + qd % 3], qd / 3), &sn, &cs);
            row[p * W + 64 + 3 + qd] = sn; row[p * W + 64 + 3 + 3 * DIR_FREQ + qd] = cs;
        }
    }
    for (int i = threadIdx.x; i < n_here * 12; i += 256) {      // the raw coordinates and the zero padding: columns 0..2, 63, 64..66, 91..95
        const int p = i / 12, k = i - 12 * p;
        const int col = k < 3 ? k : (k == 3 ? 63 : (k < 7 ? 64 + k - 4 : 91 + k - 7));
        row[p * W + col] = k < 3 ? pt[p][k] : (k >= 4 && k < 7 ? pt[p][3 + k - 4] : 0.f);
    }
    __syncthreads();
    f32x4* dst = reinterpret_cast<f32x4*>(out + base * W);      // base * 96 floats: 16-byte aligned with `out`
    for (int i = threadIdx.x; i < n_here * (W / 4); i += 256) dst[i] = reinterpret_cast<const f32x4*>(row)[i];
}

__global__ void encode_dir_kernel(RayGeom g, float* __restrict__ pe_dir) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= g.n_rays * D_DIR) return;
    const long long ray = i / D_DIR;
    const int f = (int)(i - ray * D_DIR);
    const float dx = g.rays_d[ray * 3], dy = g.rays_d[ray * 3 + 1], dz = g.rays_d[ray * 3 + 2];
    const float vx = g.m[0] * dx + g.m[1] * dy + g.m[2] * dz, vy = g.m[3] * dx + g.m[4] * dy + g.m[5] * dz,
                vz = g.m[6] * dx + g.m[7] * dy + g.m[8] * dz;
    float v;
    if (f < 3) v = pick3(vx, vy, vz, f);
    else {
        const int q = (f - 3) % (3 * DIR_FREQ);
        float sn, cs;
        pe_sincos(ldexpf(pick3(vx, vy, vz, q % 3), q / 3), &sn, &cs);
        v = (f < 3 + 3 * DIR_FREQ) ? sn : cs;
    }
    pe_dir[i] = v;
}

// ============================================================================ backward reductions
// Sum per-tile partial latent gradients, deterministic tree: every block adds up to RED_CHUNK consecutive
// tiles of one object for all n_lat*256 columns (float4 per thread, coalesced 1 KiB rows).
//   in : [obj][tiles][cols]      out : [obj][ceil(tiles/RED_CHUNK)][cols]
constexpr int RED_CHUNK = 32;
__global__ void __launch_bounds__(256) reduce_tiles_kernel(const float* __restrict__ in, long long tiles, int cols, float* __restrict__ out) {
    const long long n_chunks = (tiles + RED_CHUNK - 1) / RED_CHUNK;
    const long long chunk = blockIdx.x, obj = blockIdx.y;
    const long long t0 = chunk * RED_CHUNK;
    const long long t1 = (t0 + RED_CHUNK < tiles) ? t0 + RED_CHUNK : tiles;
    for (int c4 = threadIdx.x; c4 * 4 < cols; c4 += blockDim.x) {
        const float* p = in + (obj * tiles + t0) * cols + c4 * 4;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        long long t = t0;
        for (; t + 1 < t1; t += 2) {
            s0 += *reinterpret_cast<const f32x4*>(p);
            s1 += *reinterpret_cast<const f32x4*>(p + cols);
            p += 2 * (long long)cols;
        }
        if (t < t1) s0 += *reinterpret_cast<const f32x4*>(p);
        *reinterpret_cast<f32x4*>(out + (obj * n_chunks + chunk) * cols + c4 * 4) = s0 + s1;
    }
}

}  // namespace snr

using namespace snr;

// ============================================================================ C ABI
int snr_bf16_pack_(const float* const* W, int sb, int tb, float* packed, void* stream_);   // snr_bf16.hip

extern "C" {

int snr_abi_version(void) { return SNR_ABI_VERSION; }

static thread_local const char* g_last_err = "";
const char* snr_last_hip_error(void) { return g_last_err; }
int snr_check_launch_(void) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_err = hipGetErrorString(e); return SNR_E_LAUNCH; }
    return SNR_OK;
}

size_t snr_packed_bytes(int sb, int tb) {
    if (sb < 0 || tb < 0 || sb > MAX_BLOCKS || tb > MAX_BLOCKS) return 0;
    return (size_t)make_layout(sb, tb).total * sizeof(float);
}

size_t snr_mask_bytes(int64_t n_points, int sb, int tb) { return (size_t)mask_bytes(n_points, sb, tb); }

static inline int grid_for(long long total, int block = 256, int cap = 4096) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

int snr_pack_weights(const float* const* t, int n_tensors, int sb, int tb, float* packed, void* stream_) {
    if (!t || !packed || sb < 0 || tb < 0 || sb > MAX_BLOCKS || tb > MAX_BLOCKS) return SNR_E_ARG;
    if (n_tensors != 2 * (sb + tb + 6)) return SNR_E_SHAPE;
    for (int i = 0; i < n_tensors; ++i) if (!t[i]) return SNR_E_ARG;
    hipStream_t st = (hipStream_t)stream_;
    const Layout L = make_layout(sb, tb);
    // tensor index helpers (weight, bias pairs in reference order without the latent layers)
    int ti = 0;
    auto Wp = [&](int i) { return t[2 * i]; };
    auto Bp = [&](int i) { return t[2 * i + 1]; };
    const int i_xyz = ti++;
    int i_shape[MAX_BLOCKS]; for (int j = 0; j < sb; ++j) i_shape[j] = ti++;
    const int i_encshape = ti++;
    const int i_sigma = ti++;
    const int i_view = ti++;
    int i_tex[MAX_BLOCKS]; for (int j = 0; j < tb; ++j) i_tex[j] = ti++;
    const int i_rgb0 = ti++;
    const int i_rgb2 = ti++;
    const long long c256 = 256 * KC;

    // ---- forward stream
    float* f = packed + L.fwd;
    auto fwd = [&](const float* Wsrc, int n_out, int k_in, int n_chunks) {
        pack_fwd_kernel<<<grid_for((long long)n_chunks * n_out * KC), 256, 0, st>>>(Wsrc, n_out, k_in, n_chunks, f);
        f += (long long)n_chunks * n_out * KC;
    };
    fwd(Wp(i_xyz), 256, D_XYZ, 2);
    for (int j = 0; j < sb; ++j) fwd(Wp(i_shape[j]), 256, 256, 8);
    fwd(Wp(i_encshape), 256, 256, 8);
    fwd(Wp(i_view), 256, 256 + D_DIR, 9);
    for (int j = 0; j < tb; ++j) fwd(Wp(i_tex[j]), 256, 256, 8);
    fwd(Wp(i_rgb0), 128, 256, 8);
    if (f - (packed + L.fwd) != L.fwd_floats) return SNR_E_SHAPE;

    // ---- backward stream (reverse consumption order)
    float* b = packed + L.bwd;
    auto bwd = [&](const float* Wsrc, int n_out, int k_in, int rows_pad) {
        const int n_chunks = n_out / KC;
        pack_bwd_kernel<<<grid_for((long long)n_chunks * rows_pad * KC), 256, 0, st>>>(Wsrc, n_out, k_in, rows_pad, b);
        b += (long long)n_chunks * rows_pad * KC;
    };
    bwd(Wp(i_rgb0), 128, 256, 256);
    for (int j = tb - 1; j >= 0; --j) bwd(Wp(i_tex[j]), 256, 256, 256);
    bwd(Wp(i_view), 256, 256 + D_DIR, K_VIEW_PAD);
    bwd(Wp(i_encshape), 256, 256, 256);
    for (int j = sb - 1; j >= 0; --j) bwd(Wp(i_shape[j]), 256, 256, 256);
    bwd(Wp(i_xyz), 256, D_XYZ, K_XYZ_PAD);
    if (b - (packed + L.bwd) != L.bwd_floats) return SNR_E_SHAPE;
    (void)c256;

    // ---- vectors
    auto vec = [&](const float* src, int n, long long off, int n_pad) {
        copy_pad_kernel<<<(n_pad + 255) / 256, 256, 0, st>>>(src, n, packed + off, n_pad);
    };
    vec(Bp(i_xyz), 256, L.bias + 256ll * layer_enc_xyz(), 256);
    for (int j = 0; j < sb; ++j) vec(Bp(i_shape[j]), 256, L.bias + 256ll * layer_shape(j), 256);
    vec(Bp(i_encshape), 256, L.bias + 256ll * layer_enc_shape(sb), 256);
    vec(Bp(i_view), 256, L.bias + 256ll * layer_viewdir(sb), 256);
    for (int j = 0; j < tb; ++j) vec(Bp(i_tex[j]), 256, L.bias + 256ll * layer_texture(sb, j), 256);
    vec(Bp(i_rgb0), 128, L.bias + 256ll * layer_rgb0(sb, tb), 256);
    vec(Wp(i_sigma), 256, L.sigma_w, 256);
    vec(Bp(i_sigma), 1, L.sigma_b, 4);
    vec(Wp(i_rgb2), 3 * 128, L.rgb2_w, 3 * 128);
    vec(Bp(i_rgb2), 3, L.rgb2_b, 4);
    // ---- split-bf16 streams (same weights as hi/lo bf16 pairs, output-tile-major chunks)
    {
        const float* Wl[MAX_BLOCKS * 2 + 4];
        int n = 0;
        Wl[n++] = Wp(i_xyz);
        for (int j = 0; j < sb; ++j) Wl[n++] = Wp(i_shape[j]);
        Wl[n++] = Wp(i_encshape);
        Wl[n++] = Wp(i_view);
        for (int j = 0; j < tb; ++j) Wl[n++] = Wp(i_tex[j]);
        Wl[n++] = Wp(i_rgb0);
        int rc = snr_bf16_pack_(Wl, sb, tb, packed, stream_);
        if (rc != SNR_OK) return rc;
    }
    return snr_check_launch_();
}

int snr_composite_fwd(const float* sigmas, const float* rgbs, const float* z_vals, int z_mode, int flags, int64_t n_rays,
                      int64_t rays_per_obj, int S, float* rgb, float* depth, float* acc, void* stream_) {
    if (n_rays == 0) return SNR_OK;      /* empty ray packet: nothing to do, pointers may be null */
    if (!sigmas || !rgbs || !z_vals || !rgb || !depth || !acc) return SNR_E_ARG;
    if (n_rays < 0 || S < 1 || z_mode < 0 || z_mode > 2) return SNR_E_ARG;
    if (z_mode == SNR_Z_PER_OBJECT && (rays_per_obj < 1)) return SNR_E_SHAPE;
    if (rays_per_obj < 1) rays_per_obj = n_rays;
    const bool aligned = !(((uintptr_t)sigmas | (uintptr_t)rgbs | (uintptr_t)z_vals) & 15);
    if (S <= 64 && (S & 3) == 0 && aligned) {           /* 16-byte loads, four rays per wave */
        const int grid = grid_for(n_rays * 16, 256, 16384);
        composite_fwd16_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(sigmas, rgbs, z_vals, z_mode, flags, n_rays, rays_per_obj, S, rgb, depth, acc);
        return snr_check_launch_();
    }
    const int grid = grid_for(n_rays * 64, 256, 8192);
    composite_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(sigmas, rgbs, z_vals, z_mode, flags, n_rays, rays_per_obj, S, rgb, depth, acc);
    return snr_check_launch_();
}

int snr_scene_composite_fwd(const float* sigmas, const float* rgbs, const float* z_vals, int64_t n_pixels, int n_per_pixel, int run_length,
                            int flags, float* rgb, float* depth, float* acc, void* stream_) {
    if (n_pixels == 0) return SNR_OK;
    if (!sigmas || !rgbs || !z_vals || !rgb) return SNR_E_ARG;
    if (n_pixels < 0 || n_per_pixel < 1 || run_length < 0) return SNR_E_ARG;
    if (run_length > 0 && (n_per_pixel % run_length) != 0) return SNR_E_SHAPE;
    const size_t lds = (size_t)4 * 6 * n_per_pixel * sizeof(float);
    if (lds > 160 * 1024) return SNR_E_UNSUPPORTED;            /* more than 1706 samples per pixel */
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scene_general_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return snr_check_launch_();
    }
    const int grid = grid_for(n_pixels * 64, 256, 8192);
    hipStream_t st = (hipStream_t)stream_;
    const bool fast = n_per_pixel <= 256 && (run_length == 32 || run_length == 64 || run_length == 128);
    if (fast) {
        // two launches: the fast pass finishes the pixels it can and marks the rest, the general kernel then takes the marked ones only
        if (run_length == 32) scene_fast_kernel<32><<<grid, 256, lds, st>>>(sigmas, rgbs, z_vals, n_pixels, n_per_pixel, flags, rgb, depth, acc);
        else if (run_length == 64) scene_fast_kernel<64><<<grid, 256, lds, st>>>(sigmas, rgbs, z_vals, n_pixels, n_per_pixel, flags, rgb, depth, acc);
        else scene_fast_kernel<128><<<grid, 256, lds, st>>>(sigmas, rgbs, z_vals, n_pixels, n_per_pixel, flags, rgb, depth, acc);
        scene_general_kernel<true><<<grid, 256, lds, st>>>(sigmas, rgbs, z_vals, n_pixels, n_per_pixel, run_length, flags, rgb, depth, acc);
    } else {
        scene_general_kernel<false><<<grid, 256, lds, st>>>(sigmas, rgbs, z_vals, n_pixels, n_per_pixel, run_length, flags, rgb, depth, acc);
    }
    return snr_check_launch_();
}

int snr_composite_bwd(const float* sigmas, const float* rgbs, const float* z_vals, int z_mode, int flags, int64_t n_rays,
                      int64_t rays_per_obj, int S, const float* d_rgb, const float* d_depth, const float* d_acc,
                      float* d_sigmas, float* d_rgbs, float* d_z, void* stream_) {
    if (n_rays == 0) return SNR_OK;
    if (!sigmas || !rgbs || !z_vals || !d_sigmas || !d_rgbs) return SNR_E_ARG;
    if (n_rays < 0 || S < 1 || z_mode < 0 || z_mode > 2) return SNR_E_ARG;
    if (d_z && z_mode != SNR_Z_PER_RAY) return SNR_E_UNSUPPORTED;
    if (S > 256) return SNR_E_UNSUPPORTED;
    if (rays_per_obj < 1) rays_per_obj = n_rays;
    const int grid = grid_for(n_rays * 64, 256, 8192);
    hipStream_t st = (hipStream_t)stream_;
#define SNR_LAUNCH_CB(N) composite_bwd_kernel<N><<<grid, 256, 0, st>>>(sigmas, rgbs, z_vals, z_mode, flags, n_rays, rays_per_obj, S, \
                                                                        d_rgb, d_depth, d_acc, d_sigmas, d_rgbs, d_z)
    if (S <= 64) SNR_LAUNCH_CB(1); else if (S <= 128) SNR_LAUNCH_CB(2); else if (S <= 192) SNR_LAUNCH_CB(3); else SNR_LAUNCH_CB(4);
#undef SNR_LAUNCH_CB
    return snr_check_launch_();
}

int snr_pe_points(const float* xyz, const float* viewdir, int64_t n_points, float* out, void* stream_) {
    if (n_points == 0) return SNR_OK;
    if (!xyz || !viewdir || !out || n_points < 0 || (((uintptr_t)out) & 15)) return SNR_E_ARG;
    pe_points_kernel<<<(unsigned)((n_points + 127) / 128), 256, 0, (hipStream_t)stream_>>>(xyz, viewdir, n_points, out);
    return snr_check_launch_();
}

int snr_encode_fwd(const snr_render_args* a, float* xyz, float* viewdir, float* z_out, float* pe_xyz, float* pe_dir, uint8_t* hit, void* stream_) {
    RayGeom g;
    int rc = snr_fill_geom_(a, &g, /*need_model=*/0);
    if (rc != SNR_OK) return rc;
    if (a->n_rays == 0) return SNR_OK;
    hipStream_t st = (hipStream_t)stream_;
    const long long P = a->n_rays * a->n_samples;
    encode_kernel<<<(unsigned)((P + 255) / 256), 256, 0, st>>>(g, xyz, viewdir, z_out, pe_xyz, hit);
    if (pe_dir) encode_dir_kernel<<<(unsigned)((a->n_rays * D_DIR + 255) / 256), 256, 0, st>>>(g, pe_dir);
    return snr_check_launch_();
}

}  // extern "C"

// shared with snr_mlp.hip
int snr_fill_geom_(const snr_render_args* a, snr::RayGeom* g, int need_model) {
    if (!a || !a->rays_o || !a->rays_d) return SNR_E_ARG;
    if (a->n_rays < 0 || a->n_samples < 1 || a->z_mode < 0 || a->z_mode > SNR_Z_BOX) return SNR_E_ARG;
    if (a->z_mode == SNR_Z_BOX) {
        /* depths from the ray's own box bounds: origins are divided by z_scale, the unit-interval grid s / S is exact for powers of two only */
        if (!a->box_half || !a->z_scale) return SNR_E_ARG;
        if (a->n_samples & (a->n_samples - 1)) return SNR_E_UNSUPPORTED;
    } else if (!a->t_vals || !a->xyz_div) return SNR_E_ARG;
    if ((a->flags & SNR_METRIC_Z) && !a->z_scale) return SNR_E_ARG;
    if (a->rays_per_obj < 1 || (a->n_rays % a->rays_per_obj) != 0) return SNR_E_SHAPE;
    if (need_model) {
        if (!a->latent || !a->packed) return SNR_E_ARG;
        if (a->shape_blocks < 0 || a->texture_blocks < 0 || a->shape_blocks > snr::MAX_BLOCKS || a->texture_blocks > snr::MAX_BLOCKS)
            return SNR_E_ARG;
    }
    g->rays_o = a->rays_o; g->rays_d = a->rays_d; g->t_vals = a->t_vals; g->xyz_div = a->xyz_div; g->z_scale = a->z_scale;
    for (int i = 0; i < 9; ++i) g->m[i] = a->frame[i];
    g->xyz_mul = a->xyz_mul; g->z_mode = a->z_mode; g->flags = a->flags;
    g->n_rays = a->n_rays; g->rays_per_obj = a->rays_per_obj; g->S = a->n_samples;
    g->box_half = a->box_half; g->rng_seed = a->rng_seed; g->rng_offset = a->rng_offset; g->rng_threads = a->rng_threads;
    return SNR_OK;
}

// scratch needed behind the [obj][tiles][cols] partials for the reduction tree (floats)
long long snr_reduce_scratch_floats_(long long tiles_per_obj, int n_lat, long long n_obj) {
    const long long lvl1 = (tiles_per_obj + snr::RED_CHUNK - 1) / snr::RED_CHUNK;
    const long long lvl2 = (lvl1 + snr::RED_CHUNK - 1) / snr::RED_CHUNK;
    return (lvl1 + lvl2) * n_obj * n_lat * 256;
}

int snr_launch_reduce_latent_(const float* partial, float* scratch, long long tiles_per_obj, int n_lat, long long n_obj, float* d_latent,
                              void* stream) {
    if (n_lat <= 0 || n_obj <= 0) return SNR_OK;
    const int cols = n_lat * 256;
    const long long lvl1 = (tiles_per_obj + snr::RED_CHUNK - 1) / snr::RED_CHUNK;
    float* bufs[2] = {scratch, scratch + lvl1 * n_obj * cols};
    const float* in = partial;
    long long tiles = tiles_per_obj;
    int which = 0;
    while (true) {
        const long long n_chunks = (tiles + snr::RED_CHUNK - 1) / snr::RED_CHUNK;
        float* out = (n_chunks == 1) ? d_latent : bufs[which];
        dim3 grid((unsigned)n_chunks, (unsigned)n_obj);
        snr::reduce_tiles_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, tiles, cols, out);
        if (n_chunks == 1) break;
        in = out; tiles = n_chunks; which ^= 1;
    }
    return snr_check_launch_();
}
