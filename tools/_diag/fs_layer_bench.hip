// Micro-benchmark (development aid) of the structural alternative DESIGN.md names for the split-bf16 layer chain:
//   FEATURE-SPLIT workgroups.  A workgroup owns 128 sample points; wave w computes output features [64w, 64w+64) of every 256-wide
//   layer for ALL 128 points (2 output tiles x 4 column blocks = 8 accumulator tiles), so it needs only ITS quarter of the weights and
//   takes them straight from L2 into A-operand VGPRs (global_load_dwordx4, no LDS ring, no DMA, no fragment re-reads by other waves);
//   the activations are the B operands and live in LDS as the bf16 hi/lo operand image [plane][k16-step][column block][lane][8],
//   128 KiB, updated IN PLACE: a layer's outputs are converted and written back in four parts (one k16-step slot per wave and part),
//   part q+1 of layer l-1 under the MFMAs of phase q of layer l, which reads the slots {4w'+q}; only part 0 is exposed.
// What it measures: cycles per 256-wide layer per 128-point workgroup against the matrix floor (384 MFMAs x 32 cycles = 12.3 k) and
// against the shipped point-split kernel (16.7-19.2 k per layer, profiles/r01_v7_timelines.txt).  NL layers 256 -> 256, ReLU between,
// random weights in the shipped stream layout [layer][k16-step][tile][plane][lane][8 bf16]; workgroup 0 is checked against a host
// evaluation of the same split products.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/_diag/fs_layer_bench.hip -o tools/_diag/fs_layer_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NL = 6;
constexpr int LAYER_BYTES = 16 * 8 * 2 * 1024;       // 256 KiB: [s][tile][plane][lane][8 bf16]
constexpr int ACT_BYTES = 2 * 16 * 4 * 1024;         // 128 KiB: [plane][s][cb][lane][8 bf16]

// timing-experiment switches (outputs of these builds are garbage by design): -DNO_EPI no conversion / LDS write-back, -DNO_BAR no
// barriers, -DNO_GLOAD the A fragments are loaded once, -DIL_VALU=n VALU ops requested per MFMA (0: no interleave request)
#ifndef IL_VALU
#define IL_VALU 2
#endif
#if IL_VALU > 0
#define INTERLEAVE(N_MFMA)                                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < (N_MFMA); ++g_) {                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
        __builtin_amdgcn_sched_group_barrier(0x002, IL_VALU, 0);                           \
    }
#else
#define INTERLEAVE(N_MFMA)
#endif
#ifdef NO_BAR
#define BARRIER() do {} while (0)
#else
#define BARRIER() __syncthreads()
#endif

__host__ __device__ inline float hash_unit(unsigned a, unsigned b, unsigned c) {     // deterministic value in [-1, 1)
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

struct Frags { bf16x8 ah[2][2], al[2][2]; };     // the wave's A fragments of HALF a phase: 2 steps x 2 tiles x (hi, lo) = 32 VGPRs

// A fragments of half HH of phase Q (steps 4w'+Q, w' = 2*HH, 2*HH+1) of the layer at Wl, for this wave's tiles 2w, 2w+1
template <int Q, int HH>
__device__ __forceinline__ void load_frags(Frags& f, const char* __restrict__ Wl, int wave, int lane, bool force = false) {
#ifdef NO_GLOAD
    if (!force) { asm volatile("" : "+v"(f.ah[0][0]), "+v"(f.al[0][0])); return; }
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = 4 * (2 * HH + i) + Q;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const char* p = Wl + ((s * 8 + 2 * wave + t) * 2) * 1024 + lane * 16;
            f.ah[i][t] = *reinterpret_cast<const bf16x8*>(p);
            f.al[i][t] = *reinterpret_cast<const bf16x8*>(p + 1024);
        }
    }
}

// 48 MFMAs of half HH of phase Q: acc[t][cb] += W_t[:, step] * act[step][cb]
template <int Q, int HH>
__device__ __forceinline__ void phase_mma(f32x16 (&acc)[2][4], const Frags& f, const char* lds, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = 4 * (2 * HH + i) + Q;
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            bh[cb] = *reinterpret_cast<const bf16x8*>(lds + ((0 * 16 + s) * 4 + cb) * 1024 + lane * 16);
            bl[cb] = *reinterpret_cast<const bf16x8*>(lds + ((1 * 16 + s) * 4 + cb) * 1024 + lane * 16);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i][t], bh[cb], acc[t][cb], 0, 0, 0);
                acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i][t], bl[cb], acc[t][cb], 0, 0, 0);
                acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i][t], bh[cb], acc[t][cb], 0, 0, 0);
            }
    }
}

// HALF a part: part Q of the previous layer's outputs = tile Q>>1, registers 8*(Q&1) .. +7, column blocks 2*HH, 2*HH+1: ReLU, hi/lo split,
// into slot 4w+Q
template <int Q, int HH>
__device__ __forceinline__ void write_part(const f32x16 (&prev)[2][4], char* lds, int wave, int lane) {
#ifdef NO_EPI
    return;
#endif
    constexpr int T = Q >> 1, HALF = Q & 1;
    const int s = 4 * wave + Q;
#pragma unroll
    for (int cb = 2 * HH; cb < 2 * HH + 2; ++cb) {
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = prev[T][cb][8 * HALF + j];
            const float y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), 0));      // ReLU on the bit pattern
            const __bf16 b = (__bf16)y;
            hi[j] = b;
            lo[j] = (__bf16)(y - (float)b);
        }
        *reinterpret_cast<bf16x8*>(lds + ((0 * 16 + s) * 4 + cb) * 1024 + lane * 16) = hi;
        *reinterpret_cast<bf16x8*>(lds + ((1 * 16 + s) * 4 + cb) * 1024 + lane * 16) = lo;
    }
}


__device__ __forceinline__ void init_bias(f32x16 (&acc)[2][4], const float* __restrict__ bias, int wave, int h) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 64 * wave + 32 * t + 8 * j + 4 * h);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][cb][4 * j + e] = b[e];
        }
}

// one layer: prev (finished sums of the layer before) -> LDS in four parts under this layer's MFMAs -> cur.
// fa holds the A fragments of the first half of phase 0 on entry and those of the NEXT layer's on exit.
#ifdef VMEM_FIRST      /* ask for the 8 fragment loads of the next half-phase at the TOP of this one: LLVM otherwise sinks them to the end and the
                          wait behind the barrier exposes a whole L2 round trip */
#define VMEM_GROUP() __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
#else
#define VMEM_GROUP()
#endif
#define HALF_PHASE(Q, HH, FUSE, FNEXT_LOAD, WRITE)                                   \
    FNEXT_LOAD;                                                                      \
    phase_mma<Q, HH>(cur, FUSE, lds, lane);                                          \
    WRITE;                                                                           \
    VMEM_GROUP()                                                                     \
    INTERLEAVE(48)                                                                   \
    __builtin_amdgcn_sched_barrier(0);
__device__ __forceinline__ void layer(f32x16 (&prev)[2][4], f32x16 (&cur)[2][4], Frags& fa, Frags& fb, const char* __restrict__ Wl,
                                      const char* __restrict__ Wnext, const float* __restrict__ bias, char* lds, int wave, int lane) {
    init_bias(cur, bias, wave, lane >> 5);
    write_part<0, 0>(prev, lds, wave, lane);
    write_part<0, 1>(prev, lds, wave, lane);
    BARRIER();
    HALF_PHASE(0, 0, fa, (load_frags<0, 1>(fb, Wl, wave, lane)), (write_part<1, 0>(prev, lds, wave, lane)))
    HALF_PHASE(0, 1, fb, (load_frags<1, 0>(fa, Wl, wave, lane)), (write_part<1, 1>(prev, lds, wave, lane)))
    BARRIER();
    HALF_PHASE(1, 0, fa, (load_frags<1, 1>(fb, Wl, wave, lane)), (write_part<2, 0>(prev, lds, wave, lane)))
    HALF_PHASE(1, 1, fb, (load_frags<2, 0>(fa, Wl, wave, lane)), (write_part<2, 1>(prev, lds, wave, lane)))
    BARRIER();
    HALF_PHASE(2, 0, fa, (load_frags<2, 1>(fb, Wl, wave, lane)), (write_part<3, 0>(prev, lds, wave, lane)))
    HALF_PHASE(2, 1, fb, (load_frags<3, 0>(fa, Wl, wave, lane)), (write_part<3, 1>(prev, lds, wave, lane)))
    BARRIER();
    HALF_PHASE(3, 0, fa, (load_frags<3, 1>(fb, Wl, wave, lane)), ((void)0))
    HALF_PHASE(3, 1, fb, (load_frags<0, 0>(fa, Wnext, wave, lane)), ((void)0))
}

__global__ void __launch_bounds__(256, 1)
fs_kernel(const char* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out_sum, float* __restrict__ out_check,
          unsigned long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) char lds[ACT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, h = lane >> 5;
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    f32x16 X[2][4], Y[2][4];
    // "outputs of layer -1": the input activations in accumulator layout (feature 64w + 32t + 8(r>>2) + 4h + (r&3) of point 32cb + n)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                X[t][cb][r] = fabsf(hash_unit(blockIdx.x, 32 * cb + n, 64 * wave + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3)));
    Frags fa, fb;
    load_frags<0, 0>(fa, W, wave, lane, true);
    load_frags<0, 1>(fb, W, wave, lane, true);
#ifdef UNROLL_LAYERS
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int l = 0; l < NL; l += 2) {
        layer(X, Y, fa, fb, W + (size_t)l * LAYER_BYTES, W + (size_t)(l + 1) * LAYER_BYTES, bias + l * 256, lds, wave, lane);
        layer(Y, X, fa, fb, W + (size_t)(l + 1) * LAYER_BYTES, W + (size_t)((l + 2) % NL) * LAYER_BYTES, bias + (l + 1) * 256, lds, wave, lane);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += X[t][cb][r];
    out_sum[blockIdx.x * 256 + tid] = s;
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = t0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = t1; }
    if (blockIdx.x == 0) {      // the last layer's pre-activation sums of workgroup 0: [point][feature]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out_check[(32 * cb + n) * 256 + 64 * wave + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3)] = X[t][cb][r];
    }
}

// ---------------------------------------------------------------------------------------------- host
static inline float bf16_round(float v) {       // round to nearest even to bf16, back to float
    unsigned u; memcpy(&u, &v, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    float r; memcpy(&r, &u, 4); return r;
}
static inline unsigned short bf16_bits(float v) { unsigned u; memcpy(&u, &v, 4); return (unsigned short)(u >> 16); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n_wg = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 20;
    std::vector<float> Wf((size_t)NL * 256 * 256), bias(NL * 256);
    for (int l = 0; l < NL; ++l) {
        for (int i = 0; i < 256 * 256; ++i) Wf[(size_t)l * 65536 + i] = hash_unit(1000 + l, i / 256, i % 256) * (1.0f / 16.0f) * 1.7f;
        for (int i = 0; i < 256; ++i) bias[l * 256 + i] = hash_unit(2000 + l, i, 7) * 0.05f;
    }
    // pack: [layer][s][tile][plane][lane][8]; element j of lane (n,h) = W[row 32*tile + n][k = 16 s + 8 (j>>2) + 4 h + (j&3)]
    std::vector<unsigned short> Wp((size_t)NL * LAYER_BYTES / 2);
    std::vector<float> Whi(Wf.size()), Wlo(Wf.size());
    for (size_t i = 0; i < Wf.size(); ++i) { Whi[i] = bf16_round(Wf[i]); Wlo[i] = bf16_round(Wf[i] - Whi[i]); }
    for (int l = 0; l < NL; ++l)
        for (int s = 0; s < 16; ++s)
            for (int tile = 0; tile < 8; ++tile)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = 32 * tile + (lane & 31), k = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
                        const size_t base = (size_t)l * LAYER_BYTES / 2 + ((size_t)(s * 8 + tile) * 2) * 512 + lane * 8 + j;
                        Wp[base] = bf16_bits(Whi[(size_t)l * 65536 + row * 256 + k]);
                        Wp[base + 512] = bf16_bits(Wlo[(size_t)l * 65536 + row * 256 + k]);
                    }
    char* dW; float *dB, *dSum, *dChk; unsigned long long* dSt;
    CK(hipMalloc(&dW, Wp.size() * 2)); CK(hipMalloc(&dB, bias.size() * 4)); CK(hipMalloc(&dSum, (size_t)n_wg * 256 * 4));
    CK(hipMalloc(&dChk, 128 * 256 * 4)); CK(hipMalloc(&dSt, (size_t)n_wg * 8 * 8));
    CK(hipMemcpy(dW, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) fs_kernel<<<n_wg, 256>>>(dW, dB, dSum, dChk, dSt);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) fs_kernel<<<n_wg, 256>>>(dW, dB, dSum, dChk, dSt);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    std::vector<unsigned long long> st((size_t)n_wg * 8);
    std::vector<float> chk(128 * 256);
    CK(hipMemcpy(st.data(), dSt, st.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(chk.data(), dChk, chk.size() * 4, hipMemcpyDeviceToHost));
    double cyc = 0; for (int i = 0; i < n_wg * 4; ++i) cyc += (double)(st[2 * i + 1] - st[2 * i]);
    cyc /= (n_wg * 4);
    // host evaluation of workgroup 0 (128 points): the same split products, double accumulation
    std::vector<double> x(128 * 256), y(128 * 256);
    for (int p = 0; p < 128; ++p) for (int k = 0; k < 256; ++k) x[p * 256 + k] = fabsf(hash_unit(0, p, k));
    for (int l = 0; l < NL; ++l) {
        for (int p = 0; p < 128; ++p)
            for (int o = 0; o < 256; ++o) {
                double a = bias[l * 256 + o];
                for (int k = 0; k < 256; ++k) {
                    const float xv = (float)(x[p * 256 + k] > 0 ? x[p * 256 + k] : 0.0);
                    const float xh = bf16_round(xv), xl = bf16_round(xv - xh);
                    const double wh = Whi[(size_t)l * 65536 + o * 256 + k], wl = Wlo[(size_t)l * 65536 + o * 256 + k];
                    a += wh * xh + wh * xl + wl * xh;
                }
                y[p * 256 + o] = a;
            }
        x = y;      // (the next layer applies the ReLU when it splits)
        for (auto& v : x) v = (double)(float)v;
    }
    double worst = 0, scale = 0;
    for (int i = 0; i < 128 * 256; ++i) { worst = fmax(worst, fabs(chk[i] - y[i])); scale = fmax(scale, fabs(y[i])); }
    const double floor_cyc = 384.0 * 32.0;
    printf("feature-split layer chain: %d workgroups x 128 points, %d layers 256->256: %.4f ms/launch\n", n_wg, NL, ms);
    printf("  in-kernel: %.0f cycles per workgroup = %.0f cycles per layer (matrix floor %.0f -> %.1f %% of the pipe); shipped point-split kernel: 16.7-19.2 k per layer\n",
           cyc, cyc / NL, floor_cyc, 100.0 * floor_cyc / (cyc / NL));
    printf("  wall: %.0f workgroups per CU in sequence -> %.2f GHz effective shader clock\n", n_wg / 256.0, cyc * (n_wg / 256.0) / (ms * 1e-3) / 1e9);
    printf("  check (workgroup 0 vs host split products): max abs diff %.3e at scale %.3e -> %s\n", worst, scale, worst < 2e-4 * fmax(scale, 1.0) ? "OK" : "MISMATCH");
    return worst < 2e-4 * fmax(scale, 1.0) ? 0 : 2;
}
