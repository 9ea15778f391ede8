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
    if (KIND == 10 || KIND == 11 || KIND == 13 || KIND == 14 || KIND == 15) {
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x >> 6) * 1024u);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst) : "memory");
    }
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
            } else if (KIND == 8 || KIND == 9) {   // KIND 8: the same pieces, wave w issues one MFMA gap later than wave w-1; KIND 9: only wave 0 issues
                const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                if (KIND == 8 ? (i % 6 == wv) : (i % 6 == 0 && wv == 0)) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x >> 6) * 1024u + ((it * 2 + i / 6) & 7) * 4096u);
                    const unsigned voff = threadIdx.x * 16u + ((it * 2 + i / 6) & 15) * 4096u;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
                }
                if (i == 11 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (KIND == 10) {      // bare instruction: M0 written once before the loop
                if (i % 6 == 0) {
                    const unsigned voff = threadIdx.x * 16u + ((it * 2 + i / 6) & 15) * 4096u;
                    asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src) : "memory");
                }
                if (i == 11 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (KIND == 11) {      // M0 advanced by s_add like the kernel's ring_dma, no save/restore
                if (i % 6 == 0) {
                    const unsigned voff = threadIdx.x * 16u + ((it * 2 + i / 6) & 15) * 4096u;
                    asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_xor_b32 m0, m0, 0x1000" :: "v"(voff), "s"(src) : "memory");
                }
                if (i == 11 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (KIND == 12) {      // compiler builtin (it manages M0)
                if (i % 6 == 0) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) char*)(const char*)src + threadIdx.x * 16u + ((it * 2 + i / 6) & 15) * 4096u,
                        (__attribute__((address_space(3))) char*)lds + (threadIdx.x >> 6) * 1024u + ((it * 2 + i / 6) & 7) * 4096u, 16, 0, 0);
                }
            } else if (KIND == 13) {      // dwordx1 pieces (256 B per wave-instruction), bare
                if (i % 6 == 0) {
                    const unsigned voff = threadIdx.x * 4u + ((it * 2 + i / 6) & 15) * 4096u;
                    asm volatile("global_load_lds_dword %0, %1" :: "v"(voff), "s"(src) : "memory");
                }
                if (i == 11 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (KIND == 14) {      // bare, 64-bit VGPR address (no SGPR base), fixed M0
                if (i % 6 == 0) {
                    const char* ga = (const char*)src + threadIdx.x * 16u + ((it * 2 + i / 6) & 15) * 4096u;
                    asm volatile("global_load_lds_dwordx4 %0, off" :: "v"(ga) : "memory");
                }
                if (i == 11 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if (KIND == 15) {      // bare, SGPR base, immediate offsets -4096..3072 step 1024 over 4 iterations (M0 fixed at base + 4096)
                if (i % 6 == 0) {
                    const unsigned voff = threadIdx.x * 16u + 4096u;
                    switch ((it * 2 + i / 6) & 7) {
                    case 0: asm volatile("global_load_lds_dwordx4 %0, %1 offset:-4096" :: "v"(voff), "s"(src) : "memory"); break;
                    case 1: asm volatile("global_load_lds_dwordx4 %0, %1 offset:-3072" :: "v"(voff), "s"(src) : "memory"); break;
                    case 2: asm volatile("global_load_lds_dwordx4 %0, %1 offset:-2048" :: "v"(voff), "s"(src) : "memory"); break;
                    case 3: asm volatile("global_load_lds_dwordx4 %0, %1 offset:-1024" :: "v"(voff), "s"(src) : "memory"); break;
                    case 4: asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src) : "memory"); break;
                    case 5: asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(voff), "s"(src) : "memory"); break;
                    case 6: asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(voff), "s"(src) : "memory"); break;
                    default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff), "s"(src) : "memory"); break;
                    }
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
    printf("%-28s fillers/MFMA %d : %.2f cycles per MFMA (waves 1-3: %.2f %.2f %.2f)\n", name, NFILL, (double)h[0] / (iters * 12.0),
           (double)h[2] / (iters * 12.0), (double)h[4] / (iters * 12.0), (double)h[6] / (iters * 12.0));
}

__global__ void probe(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned buf[4096];       // 16 KiB
    for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)buf + 4096u);
    const unsigned voff = threadIdx.x * 16u + 4096u;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:-4096\n\tglobal_load_lds_dwordx4 %0, %1 offset:-1024\n\t"
                 "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(src), "s"(dst) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = buf[i];
}
static void run_probe() {
    unsigned *src, *out; static unsigned h[4096], r[4096];
    hipMalloc(&src, 16384); hipMalloc(&out, 16384);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMemcpy(src, h, 16384, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, out);
    hipMemcpy(r, out, 16384, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4096; ++i) {
        const int kb = i / 256;            // 1 KiB pieces 0 (-4096), 3 (-1024), 4 (0), 7 (+3072) of an 8 KiB window were copied
        const bool copied = kb == 0 || kb == 3 || kb == 4 || kb == 7;
        if (r[i] != (copied ? (unsigned)i : 0xdeadbeefu)) { if (bad < 8) printf("probe mismatch at dword %d: %08x\n", i, r[i]); ++bad; }
    }
    printf("immediate-offset LDS-DMA probe: %s (%d mismatches)\n", bad ? "FAIL" : "ok: offset moves the LDS and the global address together", bad);
}
int main() {
    run_probe();
    unsigned long long* out; bf16x8* src;
    hipMalloc(&out, 1024 * 8 * 8); hipMalloc(&src, 4096 * 16);
    hipMemset(src, 0x3c, 4096 * 16);
    run<0, 0>(out, src, "tile-major"); run<1, 0>(out, src, "term-major");
    run<0, 4>(out, src, "independent fma"); run<0, 5>(out, src, "independent fma"); run<0, 6>(out, src, "independent fma");
    run<0, 2, 1>(out, src, "one dependent chain"); run<0, 3, 1>(out, src, "one dependent chain"); run<0, 4, 1>(out, src, "one dependent chain");
    run<0, 4, 2>(out, src, "two chains"); run<0, 6, 2>(out, src, "two chains");
    run<0, 1, 3>(out, src, "accvgpr_read + add"); run<0, 2, 3>(out, src, "accvgpr_read + add"); run<0, 3, 3>(out, src, "accvgpr_read + add");
    run<0, 0, 5>(out, src, "LDS-DMA piece / 6 MFMA"); run<0, 0, 6>(out, src, "gload+ds_write / 6 MFMA"); run<0, 0, 7>(out, src, "gload only / 6 MFMA");
    run<0, 0, 8>(out, src, "LDS-DMA staggered by wave"); run<0, 0, 9>(out, src, "LDS-DMA wave 0 only");
    run<0, 0, 10>(out, src, "LDS-DMA bare, fixed M0"); run<0, 0, 11>(out, src, "LDS-DMA + s_xor m0"); run<0, 0, 12>(out, src, "LDS-DMA builtin"); run<0, 0, 13>(out, src, "LDS-DMA dword bare");
    run<0, 0, 14>(out, src, "LDS-DMA bare, vaddr64"); run<0, 0, 15>(out, src, "LDS-DMA bare, imm offsets");
    run<0, 0, 4>(out, src, "ds_read_b128 (8 of 12)"); run<0, 2, 4>(out, src, "ds_read_b128 + fma"); run<0, 3, 4>(out, src, "ds_read_b128 + fma"); run<0, 4, 4>(out, src, "ds_read_b128 + fma");
    return 0;
}
