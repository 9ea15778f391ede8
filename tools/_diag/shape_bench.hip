// Micro-benchmark (development aid): the split-bf16 layer chain on the two bf16 MFMA shapes of gfx950.
//   A: the SHIPPED layer body (snr::bf::layer_fwd<8,...> of csrc/snr_bf16.hip, included as it is): v_mfma_f32_32x32x16_bf16, 384 MFMAs
//      of 32 cycles per 256-wide layer and 32-point wave tile;
//   B: the same structure on v_mfma_f32_16x16x32_bf16: 768 MFMAs of 16 cycles, the wave's 32 points as two 16-point column blocks that
//      share every A fragment, the accumulators of layer l again the B operands of layer l+1 without data movement (register r of lane
//      (n, g) of tile T holds feature 16 T + 4 g + r; an operand step of 32 k takes tiles 2S and 2S+1), the same LDS-DMA ring (one 32 KiB
//      chunk = one k32-step), the same epilogue (ReLU on the bit pattern, hi / lo split) spread under the MFMAs.
// The guide (MI355X_MICROARCH.md, DVFS give-back item 7) reports that the chip holds a ~12-15 % higher clock on the 16x16x32 shape in
// MFMA-dense loops on random data at equal cycles per FLOP; this measures whether that survives the real layer body (operand delivery,
// epilogue, ring) -- wall time per launch on random data, in-kernel clock, and the two outputs checked against each other.
// NL layers 256 -> 256 with ReLU between them, 2048 workgroups x 4 waves x 32 points (the bench grid of the product kernel).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/_diag/shape_bench.hip -o tools/_diag/shape_bench
#include "../../sup-nerf_amd/csrc/snr_bf16.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int snr_check_launch_() { return hipGetLastError() == hipSuccess ? 0 : -4; }      // (the library's helper lives in snr_aux.hip)

namespace sb {
using namespace snr;
using namespace snr::bf;

constexpr int NL = 6;
constexpr int LAYER_BYTES = 256 * 1024;

__host__ __device__ inline float hash_unit(unsigned a, unsigned b, unsigned c) {     // deterministic value in [-1, 1)
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}
__host__ __device__ inline float weight_of(int layer, int row, int k) { return hash_unit(layer + 1, row, k) * 0.108f; }   // ~ sqrt(6 / 512)

// ---- packing: A = the shipped image [k16-step][tile32][plane][lane][8], B = [k32-step][tile16][plane][lane][8]
__global__ void pack_a(__bf16* dst) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;       // over NL * 16 * 8 * 64 * 8
    if (i >= (long long)NL * 16 * 8 * 512) return;
    const int j = i & 7, lane = (i >> 3) & 63, tile = (i >> 9) & 7, s = (i >> 12) & 15, layer = (int)(i >> 16);
    const int row = 32 * tile + (lane & 31), k = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
    const float v = weight_of(layer, row, k);
    const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
    const long long base = (long long)layer * (LAYER_BYTES / 2) + (((long long)s * 8 + tile) * 2) * 512;
    dst[base + lane * 8 + j] = hi; dst[base + 512 + lane * 8 + j] = lo;
}
__global__ void pack_b(__bf16* dst) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;       // over NL * 8 * 16 * 64 * 8
    if (i >= (long long)NL * 8 * 16 * 512) return;
    const int j = i & 7, lane = (i >> 3) & 63, tile = (i >> 9) & 15, s = (i >> 13) & 7, layer = (int)(i >> 16);
    const int row = 16 * tile + (lane & 15), k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    const float v = weight_of(layer, row, k);
    const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
    const long long base = (long long)layer * (LAYER_BYTES / 2) + (((long long)s * 16 + tile) * 2) * 512;
    dst[base + lane * 8 + j] = hi; dst[base + 512 + lane * 8 + j] = lo;
}

__device__ __forceinline__ void stage_zeros(char* lds, int tid) {
    float* vec = reinterpret_cast<float*>(lds + OFF_VEC);
    for (int i = tid; i < VEC_FLOATS; i += 256) vec[i] = 0.f;
    __syncthreads();
}
__device__ __forceinline__ void stamp(unsigned long long* st, int slot) {
    if (st && threadIdx.x == 0) {
        unsigned long long c, r;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r) :: "memory");
        st[blockIdx.x * 4 + 2 * slot] = c; st[blockIdx.x * 4 + 2 * slot + 1] = r;
    }
}

// ------------------------------------------------------------------------------------------ A: the shipped layer body
__global__ void __launch_bounds__(256, 1) chain_a(const char* __restrict__ stream, float* __restrict__ out, unsigned long long* st) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5;
    const long long tile32 = blockIdx.x * 4ll + wave;
    stage_zeros(lds, tid);
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = hash_unit(0, (unsigned)(tile32 * 32 + p), 32 * t + 8 * (r >> 2) + 4 * h + (r & 3));
    stamp(st, 0);
    Ring ring;
    const unsigned voff = lane * 16u + 4096u;
    ring_start(ring, stream, NL * 8, lds, voff);
    XOp x[16];
    uint32_t mask[4];
    const float* vec = reinterpret_cast<const float*>(lds + OFF_VEC);
    FwdEpi c{0, vec + VEC_BIAS, vec + VEC_ZERO, nullptr};
#pragma unroll 1
    for (int l = 0; l < NL; ++l) layer_fwd<8, false, false, false>(acc, x, nullptr, ring, lds, c, false, mask, tid, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(st, 1);
    if (out) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(tile32 * 32 + p) * 256 + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3)] = acc[t][r];
    }
}

// ------------------------------------------------------------------------------------------ B: 16x16x32
#ifndef ILB16_VALU
#define ILB16_VALU 1
#endif
#ifndef ILB16_MFMA
#define ILB16_MFMA 2
#endif
#if ILB16_VALU > 0
#define IL16(N_MFMA)                                                                     \
    _Pragma("unroll") for (int g_ = 0; g_ < (N_MFMA) * 2 / ILB16_MFMA; ++g_) {           \
        __builtin_amdgcn_sched_group_barrier(0x008, ILB16_MFMA, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x002, ILB16_VALU, 0);                      \
    }
#else
#define IL16(N_MFMA)
#endif

struct Frag16 { bf16x8 hi[4], lo[4]; };        // A fragments of four 16-row tiles
template <int T0>
__device__ __forceinline__ void load16(Frag16& f, const char* ws) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f.hi[t] = *reinterpret_cast<const bf16x8*>(ws + (2 * (T0 + t)) * 1024);
        f.lo[t] = *reinterpret_cast<const bf16x8*>(ws + (2 * (T0 + t) + 1) * 1024);
    }
}
// four values of one finished 16x16 tile -> elements 4*HALF .. 4*HALF+3 of an operand step (ReLU on the bit pattern, hi / lo split)
template <int HALF>
__device__ __forceinline__ void epi16(const f32x4& a, XOp& o) {
#ifdef SB_NOEPI      /* timing only: no conversion */
    return;
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = a[e];
        const float y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0));
        split_store(y, o, 4 * HALF + e);
    }
    if (HALF == 1) pin(o);
}
template <int T0, bool TO_P, bool FIRST = false>
__device__ __forceinline__ void mma16(f32x4 (&accC)[2][16], f32x4 (&accP)[2][16], const XOp (&x)[2], const Frag16& f, const float* bias = nullptr, int g = 0) {
#ifdef SB_BIAS_C     /* the layer's first step starts every accumulation chain from the bias held in VGPRs (the MFMA's C operand): no accumulator writes */
    f32x4 bv[4];
    if constexpr (FIRST) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = *reinterpret_cast<const f32x4*>(bias + 16 * (T0 + t) + 4 * g);
    }
#endif
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#ifdef SB_BIAS_C
            f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.hi[t], x[c].hi, FIRST ? bv[t] : accC[c][T0 + t], 0, 0, 0);
#else
            f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.hi[t], x[c].hi, accC[c][T0 + t], 0, 0, 0);
#endif
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.hi[t], x[c].lo, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.lo[t], x[c].hi, a, 0, 0, 0);
            if (TO_P) accP[c][T0 + t] = a; else accC[c][T0 + t] = a;
        }
}

// One 256 -> 256 layer: accP = the previous layer's finished accumulators (in), this layer's (out).  `w` = chunk of k32-step 0, already
// acquired, with the fragments of its first four tiles in `fa`; on return the same holds for the NEXT layer's step 0.
__device__ __forceinline__ void layer16(f32x4 (&accP)[2][16], Ring& ring, char* lds, const float* __restrict__ bias, const char*& w, Frag16& fa,
                                        int lane) {
    f32x4 accC[2][16];
    const int g = lane >> 4;
    const unsigned voff = lane * 16u + 4096u;
#ifndef SB_BIAS_C
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * g);
        accC[0][t] = b; accC[1][t] = b;
    }
#endif
    XOp xc[2], xn[2];
    epi16<0>(accP[0][0], xc[0]); epi16<1>(accP[0][1], xc[0]);
    epi16<0>(accP[1][0], xc[1]); epi16<1>(accP[1][1], xc[1]);
    Frag16 fb;
#define SB_STEP(S)                                                                                                          \
    {                                                                                                                       \
        constexpr bool LASTS = (S) == 7;                                                                                    \
        /* group 0: tiles 0..3 (fragments in fa), fetch tiles 4..7 */                                                       \
        load16<4>(fb, w);                                                                                                   \
        mma16<0, LASTS, (S) == 0>(accC, accP, xc, fa, bias, g);                                                                              \
        ring_pieces<2, 2>(ring, voff);                                                                                      \
        if constexpr (!LASTS) epi16<0>(accP[0][2 * ((S) + 1)], xn[0]);                                                       \
        IL16(12) __builtin_amdgcn_sched_barrier(0);                                                                         \
        /* group 1 */                                                                                                       \
        load16<8>(fa, w);                                                                                                   \
        mma16<4, LASTS, (S) == 0>(accC, accP, xc, fb, bias, g);                                                                              \
        ring_pieces<4, 2>(ring, voff);                                                                                      \
        if constexpr (!LASTS) epi16<1>(accP[0][2 * ((S) + 1) + 1], xn[0]);                                                   \
        IL16(12) __builtin_amdgcn_sched_barrier(0);                                                                         \
        /* group 2 */                                                                                                       \
        load16<12>(fb, w);                                                                                                  \
        mma16<8, LASTS, (S) == 0>(accC, accP, xc, fa, bias, g);                                                                              \
        ring_pieces<6, 2>(ring, voff);                                                                                      \
        if constexpr (!LASTS) epi16<0>(accP[1][2 * ((S) + 1)], xn[1]);                                                       \
        IL16(12) __builtin_amdgcn_sched_barrier(0);                                                                         \
        /* group 3: acquire the next step's chunk, its first fragments, the first two pieces of the chunk after it */      \
        w = ring_acquire(ring, lds) + lane * 16;                                                                            \
        load16<0>(fa, w);                                                                                                   \
        mma16<12, LASTS, (S) == 0>(accC, accP, xc, fb, bias, g);                                                                              \
        ring_pieces<0, 2>(ring, voff);                                                                                      \
        if constexpr (!LASTS) epi16<1>(accP[1][2 * ((S) + 1) + 1], xn[1]);                                                   \
        IL16(12) __builtin_amdgcn_sched_barrier(0);                                                                         \
        if constexpr (!LASTS) { xc[0] = xn[0]; xc[1] = xn[1]; }                                                             \
    }
    SB_STEP(0) SB_STEP(1) SB_STEP(2) SB_STEP(3) SB_STEP(4) SB_STEP(5) SB_STEP(6) SB_STEP(7)
#undef SB_STEP
}

__global__ void __launch_bounds__(256, 1) chain_b(const char* __restrict__ stream, float* __restrict__ out, unsigned long long* st) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const long long tile32 = blockIdx.x * 4ll + wave;
    stage_zeros(lds, tid);
    f32x4 acc[2][16];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][t][r] = hash_unit(0, (unsigned)(tile32 * 32 + 16 * c + n), 16 * t + 4 * g + r);
    stamp(st, 0);
    Ring ring;
    const unsigned voff = lane * 16u + 4096u;
    ring_start(ring, stream, NL * 8, lds, voff);
    const char* w = ring_acquire(ring, lds) + lane * 16;
    Frag16 fa;
    load16<0>(fa, w);
    ring_pieces<0, 2>(ring, voff);
    const float* vec = reinterpret_cast<const float*>(lds + OFF_VEC);
#pragma unroll 1
    for (int l = 0; l < NL; ++l) layer16(acc, ring, lds, vec + VEC_BIAS, w, fa, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" :: "v"(fa.hi[0]), "v"(fa.lo[0]));
    stamp(st, 1);
    if (out) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(tile32 * 32 + 16 * c + n) * 256 + 16 * t + 4 * g + r] = acc[c][t][r];
    }
}
}  // namespace sb

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    using namespace sb;
    const int wgs = argc > 1 ? atoi(argv[1]) : 2048, rounds = argc > 2 ? atoi(argv[2]) : 6, reps = 20;
    const long long P = wgs * 128ll;
    char *sa, *sb_;
    float *oa, *ob;
    unsigned long long* st;
    CK(hipMalloc(&sa, (size_t)NL * LAYER_BYTES)); CK(hipMalloc(&sb_, (size_t)NL * LAYER_BYTES));
    CK(hipMalloc(&oa, P * 256 * 4)); CK(hipMalloc(&ob, P * 256 * 4)); CK(hipMalloc(&st, wgs * 4 * 8));
    pack_a<<<(NL * 16 * 8 * 512 + 255) / 256, 256>>>(reinterpret_cast<__bf16*>(sa));
    pack_b<<<(NL * 8 * 16 * 512 + 255) / 256, 256>>>(reinterpret_cast<__bf16*>(sb_));
    chain_a<<<wgs, 256>>>(sa, oa, nullptr);
    chain_b<<<wgs, 256>>>(sb_, ob, nullptr);
    CK(hipDeviceSynchronize());
    {   // the two shapes compute the same split products (in a different summation order): compare
        std::vector<float> ha(P * 256), hb(P * 256);
        CK(hipMemcpy(ha.data(), oa, P * 256 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), ob, P * 256 * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; long long nz = 0;
        for (long long i = 0; i < P * 256; ++i) { md = fmax(md, fabs((double)ha[i] - hb[i])); mx = fmax(mx, fabs((double)ha[i])); nz += ha[i] != 0.f; }
        printf("outputs after %d layers: max |A| %.4f, non-zero %.1f %%, max |A - B| %.3e\n", NL, mx, 100.0 * nz / (P * 256.0), md);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](int which) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) { if (which == 0) chain_a<<<wgs, 256>>>(sa, nullptr, nullptr); else chain_b<<<wgs, 256>>>(sb_, nullptr, nullptr); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
    };
    for (int i = 0; i < 40; ++i) { chain_a<<<wgs, 256>>>(sa, nullptr, nullptr); chain_b<<<wgs, 256>>>(sb_, nullptr, nullptr); }      // warm clocks
    float best[2] = {1e9f, 1e9f};
    for (int r = 0; r < rounds; ++r)
        for (int which = 0; which < 2; ++which) {
            const float ms = time(which);
            best[which] = fminf(best[which], ms);
            printf("round %d %s: %.4f ms per launch (%d layers, %d workgroups)\n", r, which ? "B 16x16x32" : "A 32x32x16", ms, NL, wgs);
        }
    for (int which = 0; which < 2; ++which) {      // in-kernel clock and cycles per layer (stamped launches, after the timing)
        for (int i = 0; i < 5; ++i) { if (which == 0) chain_a<<<wgs, 256>>>(sa, nullptr, st); else chain_b<<<wgs, 256>>>(sb_, nullptr, st); }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(wgs * 4);
        CK(hipMemcpy(h.data(), st, wgs * 32, hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk;
        for (int b = 0; b < wgs; ++b) { const double dc = (double)(h[b * 4 + 2] - h[b * 4]), dr = (double)(h[b * 4 + 3] - h[b * 4 + 1]); cyc.push_back(dc); clk.push_back(dc / dr * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        printf("%s: best %.4f ms; median %.0f shader cycles per workgroup = %.0f per layer (matrix floor 12288), in-kernel clock %.2f GHz\n",
               which ? "B 16x16x32" : "A 32x32x16", best[which], cyc[wgs / 2], cyc[wgs / 2] / NL, clk[wgs / 2]);
    }
    printf("B / A wall: %.3f\n", best[1] / best[0]);
    return 0;
}
