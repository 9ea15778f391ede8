// Micro-benchmark (development aid): cycles per v_mfma_f32_32x32x16_bf16 for one wave per SIMD under the operand
// patterns the fused kernel uses.  Build: hipcc -O3 --offload-arch=gfx950 mfma_bench.hip -o mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VARIANT>
__global__ void __launch_bounds__(256, 1) k(unsigned long long* out, int iters, const bf16x8* src) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) reinterpret_cast<bf16x8*>(lds)[i] = src[i & 255];
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 xh = src[lane], xl = src[64 + lane];
    float filler[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    f32x16 acc2[8];                       // a second, finished accumulator set (lives in AGPRs like the kernel's accP)
    for (int t = 0; t < 8; ++t) { for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f; acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, xl, acc2[t], 0, 0, 0); }
    const char* w = lds + lane * 16;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) {           // pure MFMA, operands in registers
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, xl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, xh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, xh, acc[t], 0, 0, 0);
            }
        } else {                       // A fragments from LDS (16 x ds_read_b128 per 24 MFMAs)
            bf16x8 fh[8], fl[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                fh[t] = *reinterpret_cast<const bf16x8*>(w + ((it & 1) * 16 + 2 * t) * 1024);
                fl[t] = *reinterpret_cast<const bf16x8*>(w + ((it & 1) * 16 + 2 * t + 1) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[t], xh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[t], xl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[t], xh, acc[t], 0, 0, 0);
                if (VARIANT == 2) {    // + 6 VALU per tile
#pragma unroll
                    for (int q = 0; q < 6; ++q) filler[q] = filler[q] * 1.0001f + 0.5f;
                }
                if (VARIANT == 3) {    // + 6 VALU per tile whose inputs are read out of the other accumulator set
#pragma unroll
                    for (int q = 0; q < 6; ++q) filler[q] = filler[q] * 1.0001f + acc2[t][q + 6 * (it & 1)];
                }
                if (VARIANT == 4) {    // the kernel's epilogue op mix on one value per MFMA (read, add, med3, add, cvt/split)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float a = acc2[t][q + 2 * (it & 7)] + filler[q];
                        a = __builtin_amdgcn_fmed3f(a, 0.f, __builtin_inff()) + filler[q + 2];
                        __bf16 hi = (__bf16)a; filler[q + 4] += (float)(__bf16)(a - (float)hi) + (float)hi;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t][lane & 15];
    for (int q = 0; q < 8; ++q) s += filler[q];
    for (int t = 0; t < 8; ++t) s += acc2[t][lane & 15];
    if (lane == 0) { out[blockIdx.x * 8 + (threadIdx.x >> 6) * 2] = t1 - t0; out[blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + 1] = (unsigned long long)s; }
}

int main() {
    unsigned long long* out; bf16x8* src;
    hipMalloc(&out, 1024 * 8 * 8); hipMalloc(&src, 4096 * 16);
    hipMemset(src, 0x3c, 4096 * 16);
    const int iters = 2000, grid = 256;
    for (int v = 0; v < 5; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v == 0) k<0><<<grid, 256>>>(out, iters, src);
            if (v == 1) k<1><<<grid, 256>>>(out, iters, src);
            if (v == 2) k<2><<<grid, 256>>>(out, iters, src);
            if (v == 3) k<3><<<grid, 256>>>(out, iters, src);
            if (v == 4) k<4><<<grid, 256>>>(out, iters, src);
            hipDeviceSynchronize();
        }
        unsigned long long h[8];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d: %.1f cycles per MFMA (wave 0 of block 0; %d iters x 24)\n", v, (double)h[0] / (iters * 24.0), iters);
    }
    return 0;
}
