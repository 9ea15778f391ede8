// Micro-benchmark (development aid): does the ORDER of the three split products matter when VALU work sits in the MFMA
// gaps?  tile-major = hh(t) hl(t) lh(t) back to back on one accumulator; term-major = hh(0..3) hl(0..3) lh(0..3).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_order.hip -o mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// FILL kinds: 0 independent v_fma, 1 one dependent chain of v_fma, 2 two chains, 3 v_accvgpr_read (+ add), 4 ds_read_b128 + independent fma
template <int ORDER, int NFILL, int KIND = 0>
__global__ void __launch_bounds__(256, 1) k(unsigned long long* out, int iters, const bf16x8* src) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 xh = src[lane], xl = src[64 + lane];
    bf16x8 fh[4], fl[4];
    for (int t = 0; t < 4; ++t) { fh[t] = src[128 + 64 * t + lane]; fl[t] = src[512 + 64 * t + lane]; }
    float f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    __shared__ __attribute__((aligned(16))) char lds[32 * 1024];
    for (int i = threadIdx.x; i < 32 * 1024 / 16; i += 256) reinterpret_cast<bf16x8*>(lds)[i] = src[i & 255];
    __syncthreads();
    f32x16 acc2[4];
    for (int t = 0; t < 4; ++t) { for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f; acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, xl, acc2[t], 0, 0, 0); asm volatile("" : "+a"(acc2[t])); }
    bf16x8 dd[4];
    for (int t = 0; t < 4; ++t) dd[t] = xh;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int t = ORDER == 0 ? i / 3 : i % 4, kk = ORDER == 0 ? i % 3 : i / 4;
            if (kk == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[t], xh, acc[t], 0, 0, 0);
            else if (kk == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[t], xl, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[t], xh, acc[t], 0, 0, 0);
            if (KIND == 0) {
#pragma unroll
                for (int q = 0; q < NFILL; ++q) { f[q] = f[q] * 1.0001f + 0.5f; asm volatile("" : "+v"(f[q])); }
            } else if (KIND == 1) {
#pragma unroll
                for (int q = 0; q < NFILL; ++q) { f[0] = f[0] * 1.0001f + 0.5f; asm volatile("" : "+v"(f[0])); }
            } else if (KIND == 2) {
#pragma unroll
                for (int q = 0; q < NFILL; ++q) { f[q & 1] = f[q & 1] * 1.0001f + 0.5f; asm volatile("" : "+v"(f[q & 1])); }
            } else if (KIND == 3) {
#pragma unroll
                for (int q = 0; q < NFILL; ++q) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc2[i & 3][q])); f[q] += v; asm volatile("" : "+v"(f[q])); }
            } else if (KIND == 5) {       // one LDS-DMA piece (1 KiB per wave) every 6 MFMAs = 8 per 48, as the weight ring issues them
                if (i % 6 == 0) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x >> 6) * 1024u + ((it * 2 + i / 6) & 7) * 4096u);
                    const unsigned voff = threadIdx.x * 16u + ((it * 2 + i / 6) & 15) * 4096u;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
                }
                if (i == 11 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (KIND == 6) {       // the same bytes through registers: global_load_dwordx4 now, ds_write_b128 of the previous one
                if (i % 6 == 0) {
                    *reinterpret_cast<bf16x8*>(lds + (threadIdx.x >> 6) * 1024 + ((it * 2 + i / 6) & 7) * 4096 + lane * 16) = dd[(i / 6) & 1];
                    dd[(i / 6) & 1] = src[(threadIdx.x + ((it * 2 + i / 6) & 15) * 256) & 4095];
                }
            } else if (KIND == 7) {       // only the register loads
                if (i % 6 == 0) dd[(i / 6) & 1] = src[(threadIdx.x + ((it * 2 + i / 6) & 15) * 256) & 4095];
                if (i == 11) asm volatile("" : "+v"(dd[0]), "+v"(dd[1]));
            } else if (KIND == 4) {
                if (i < 8) { dd[i & 3] = *reinterpret_cast<const bf16x8*>(lds + (it & 1) * 16384 + i * 1024 + lane * 16); }
#pragma unroll
                for (int q = 0; q < NFILL; ++q) { f[q] = f[q] * 1.0001f + 0.5f; asm volatile("" : "+v"(f[q])); }
                if (i == 11) { asm volatile("" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3])); }
            }
            asm volatile("" : "+a"(acc[t]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0;
    for (int t = 0; t < 4; ++t) s += acc[t][lane & 15];
    for (int q = 0; q < 8; ++q) s += f[q];
    if (lane == 0) { out[blockIdx.x * 8 + (threadIdx.x >> 6) * 2] = t1 - t0; out[blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + 1] = (unsigned long long)s; }
}

template <int ORDER, int NFILL, int KIND = 0>
void run(unsigned long long* out, const bf16x8* src, const char* name) {
    const int iters = 2000, grid = 256;
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) k<ORDER, NFILL, KIND><<<grid, 256>>>(out, iters, src);
    hipDeviceSynchronize();
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-28s fillers/MFMA %d : %.2f cycles per MFMA\n", name, NFILL, (double)h[0] / (iters * 12.0));
}

int main() {
    unsigned long long* out; bf16x8* src;
    hipMalloc(&out, 1024 * 8 * 8); hipMalloc(&src, 4096 * 16);
    hipMemset(src, 0x3c, 4096 * 16);
    run<0, 0>(out, src, "tile-major"); run<1, 0>(out, src, "term-major");
    run<0, 4>(out, src, "independent fma"); run<0, 5>(out, src, "independent fma"); run<0, 6>(out, src, "independent fma");
    run<0, 2, 1>(out, src, "one dependent chain"); run<0, 3, 1>(out, src, "one dependent chain"); run<0, 4, 1>(out, src, "one dependent chain");
    run<0, 4, 2>(out, src, "two chains"); run<0, 6, 2>(out, src, "two chains");
    run<0, 1, 3>(out, src, "accvgpr_read + add"); run<0, 2, 3>(out, src, "accvgpr_read + add"); run<0, 3, 3>(out, src, "accvgpr_read + add");
    run<0, 0, 5>(out, src, "LDS-DMA piece / 6 MFMA"); run<0, 0, 6>(out, src, "gload+ds_write / 6 MFMA"); run<0, 0, 7>(out, src, "gload only / 6 MFMA");
    run<0, 0, 4>(out, src, "ds_read_b128 (8 of 12)"); run<0, 2, 4>(out, src, "ds_read_b128 + fma"); run<0, 3, 4>(out, src, "ds_read_b128 + fma"); run<0, 4, 4>(out, src, "ds_read_b128 + fma");
    return 0;
}
