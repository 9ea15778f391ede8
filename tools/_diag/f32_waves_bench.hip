// Micro-benchmark (development aid): what the fp32 MFMA pipe of ONE SIMD delivers under the issue patterns of the fused fp32
// decoder -- v_mfma_f32_32x32x2_f32 with one 512-register wave per SIMD (snr_mlp.hip) against v_mfma_f32_16x16x4_f32 with one or two waves
// per SIMD (snr_mlp16.hip) -- as cycles of SIMD time per 64 FLOP/clk "slot" (= 32 cycles of a 16x16x4, 64 of a 32x32x2), in-kernel
// s_memtime over a long loop, median over workgroups, all CUs busy.
// Build: hipcc -O3 --offload-arch=gfx950 tools/_diag/f32_waves_bench.hip -o tools/_diag/f32_waves_bench ; run: tools/_diag/f32_waves_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xc07f)

// V: 0 bare MFMAs, operands in registers (pairs of accumulators interleaved)
//    1 + A fragments from LDS, requested a group (8 MFMAs) ahead, explicit wait at the top of the group (the kernel's tile_mma)
//    2 + five VALU per second group (the epilogue slice)
//    3 + an s_barrier every 128 MFMAs per wave (the chunk rendezvous), no DMA
//    4 + the LDS-DMA pieces of the weight ring (32 KiB per 128 MFMAs and wave set), vmcnt(0) before the barrier
//    5 = 4 with waves >= WAVES/2 delayed by s_sleep after every barrier (the stagger)
template <int WAVES, int V>
__global__ void __launch_bounds__(WAVES * 64, (WAVES * 64) / 256) k16(unsigned long long* out, int chunks, const float* src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 8192; i += WAVES * 64) lds[i] = src[i & 4095];
    __syncthreads();
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 x = {src[lane], src[64 + lane], src[128 + lane], src[192 + lane]};
    float fill[4] = {1.f, 2.f, 3.f, 4.f};
    const float* wrow = lds + (lane & 15) * 32 + (lane >> 4) * 4;
    int cur = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int c = 0; c < chunks; ++c) {
        const float* wb = wrow + cur * 8192;
        if (V >= 5 && wave >= WAVES / 2) __builtin_amdgcn_s_sleep(2);       // 2 x 64 cycles
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
            f32x4 a0, a1;
            if (V >= 1) { a0 = *reinterpret_cast<const f32x4*>(wb + tile * 16); a1 = *reinterpret_cast<const f32x4*>(wb + tile * 16 + 512); }
            else { a0 = x; a1 = x; }
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                f32x4 n0 = a0, n1 = a1;
                if (V >= 1) {
                    WAIT_LDS();
                    if (t + 2 < 16) { n0 = *reinterpret_cast<const f32x4*>(wb + tile * 16 + (t + 2) * 512); n1 = *reinterpret_cast<const f32x4*>(wb + tile * 16 + (t + 3) * 512); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], x[r], acc[t], 0, 0, 0);
                    acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], x[r], acc[t + 1], 0, 0, 0);
                }
                if (V >= 2 && (t & 2)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) fill[q] = fill[q] * 1.0001f + 0.5f;
                    fill[0] += fill[3];
                }
                if (V >= 4 && tile == 0 && (WAVES == 8 ? ((t & 2) == 0) : true)) {
                    typedef const __attribute__((address_space(1))) void* gptr_t;
                    typedef __attribute__((address_space(3))) void* lptr_t;
                    const int i = WAVES == 8 ? t / 4 : t / 2;
                    const int wv = __builtin_amdgcn_readfirstlane(wave);
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + ((size_t)(c & 7) * 8192 + (i * WAVES * 64 + tid) * 4) % (1 << 20)),
                                                     (lptr_t)(lds + (cur ^ 1) * 8192 + (i * WAVES * 64 + wv * 64) * 4), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                a0 = n0; a1 = n1;
            }
        }
        if (V >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (V >= 3) __syncthreads();
        if (V >= 4) cur ^= 1;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = fill[0] + fill[1] + fill[2] + fill[3];
    for (int t = 0; t < 16; ++t) s += acc[t][lane & 3];
    if (lane == 0) { out[(blockIdx.x * WAVES + wave) * 2] = t1 - t0; out[(blockIdx.x * WAVES + wave) * 2 + 1] = (unsigned long long)s; }
}

// the 32x32x2 form, one wave per SIMD: 8 accumulator tiles, A fragments from LDS (one ds_read_b128 per 4 MFMAs), V as above (0, 1, 3, 4)
template <int V>
__global__ void __launch_bounds__(256, 1) k32(unsigned long long* out, int chunks, const float* src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 8192; i += 256) lds[i] = src[i & 4095];
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float b[16];
    for (int r = 0; r < 16; ++r) b[r] = src[r * 64 + lane];
    const float* wrow = lds + (lane & 31) * 32 + (lane >> 5) * 4;
    int cur = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int c = 0; c < chunks; ++c) {
        const float* wb = wrow + cur * 8192;
        if (V >= 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                typedef const __attribute__((address_space(1))) void* gptr_t;
                typedef __attribute__((address_space(3))) void* lptr_t;
                const int wv = __builtin_amdgcn_readfirstlane(wave);
                __builtin_amdgcn_global_load_lds((gptr_t)(src + ((size_t)(c & 7) * 8192 + (i * 256 + tid) * 4) % (1 << 20)),
                                                 (lptr_t)(lds + (cur ^ 1) * 8192 + (i * 256 + wv * 64) * 4), 16, 0, 0);
            }
        }
        f32x4 a = V >= 1 ? *reinterpret_cast<const f32x4*>(wb) : f32x4{b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                f32x4 an = a;
                if (V >= 1) {
                    if (t + 1 < 8) an = *reinterpret_cast<const f32x4*>(wb + 8 * j + (t + 1) * 1024);
                    else if (j + 1 < 4) an = *reinterpret_cast<const f32x4*>(wb + 8 * (j + 1));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[4 * j + e], acc[t], 0, 0, 0);
                a = an;
            }
        if (V >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (V >= 3) __syncthreads();
        if (V >= 4) cur ^= 1;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][lane & 15];
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = (unsigned long long)s; }
}

static double median_cycles(unsigned long long* dout, int n);

// The same work as k16<WAVES, 4> (fragments from LDS, epilogue VALU, DMA ring, barrier per chunk) with the non-matrix instructions SPREAD: at most
// one small item behind every MFMA (fragment request behind MFMAs 0 and 1 of a group, one or two VALU behind 2..5, the DMA piece behind 5),
// every position pinned, instead of one lump behind the group's eight MFMAs.  FINE = 1: as described; FINE = 2: without the barrier / DMA
// (compare with V2).
template <int WAVES, int FINE>
__global__ void __launch_bounds__(WAVES * 64, (WAVES * 64) / 256) k16f(unsigned long long* out, int chunks, const float* src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 8192; i += WAVES * 64) lds[i] = src[i & 4095];
    __syncthreads();
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 x = {src[lane], src[64 + lane], src[128 + lane], src[192 + lane]};
    float fill[4] = {1.f, 2.f, 3.f, 4.f};
    const float* wrow = lds + (lane & 15) * 32 + (lane >> 4) * 4;
    int cur = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#define SB() __builtin_amdgcn_sched_barrier(0)
    for (int c = 0; c < chunks; ++c) {
        const float* wb = wrow + cur * 8192;
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
            f32x4 a0 = *reinterpret_cast<const f32x4*>(wb + tile * 16), a1 = *reinterpret_cast<const f32x4*>(wb + tile * 16 + 512);
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                f32x4 n0 = a0, n1 = a1;
                WAIT_LDS();
                SB();
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], x[0], acc[t], 0, 0, 0);
                if (t + 2 < 16) n0 = *reinterpret_cast<const f32x4*>(wb + tile * 16 + (t + 2) * 512);
                SB();
                acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], x[0], acc[t + 1], 0, 0, 0);
                if (t + 2 < 16) n1 = *reinterpret_cast<const f32x4*>(wb + tile * 16 + (t + 3) * 512);
                SB();
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], x[1], acc[t], 0, 0, 0);
                if (t & 2) { fill[0] = fill[0] * 1.0001f + 0.5f; fill[1] = fill[1] * 1.0001f + 0.5f; }
                SB();
                acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], x[1], acc[t + 1], 0, 0, 0);
                if (t & 2) { fill[2] = fill[2] * 1.0001f + 0.5f; fill[3] = fill[3] * 1.0001f + 0.5f; }
                SB();
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2], x[2], acc[t], 0, 0, 0);
                if (t & 2) fill[0] += fill[3];
                SB();
                acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[2], x[2], acc[t + 1], 0, 0, 0);
                if (FINE == 1 && tile == 0 && (WAVES == 8 ? ((t & 2) == 0) : true)) {
                    typedef const __attribute__((address_space(1))) void* gptr_t;
                    typedef __attribute__((address_space(3))) void* lptr_t;
                    const int i = WAVES == 8 ? t / 4 : t / 2;
                    const int wv = __builtin_amdgcn_readfirstlane(wave);
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + ((size_t)(c & 7) * 8192 + (i * WAVES * 64 + tid) * 4) % (1 << 20)),
                                                     (lptr_t)(lds + (cur ^ 1) * 8192 + (i * WAVES * 64 + wv * 64) * 4), 16, 0, 0);
                }
                SB();
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[3], x[3], acc[t], 0, 0, 0);
                SB();
                acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[3], x[3], acc[t + 1], 0, 0, 0);
                SB();
                a0 = n0; a1 = n1;
            }
        }
        if (FINE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); cur ^= 1; }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = fill[0] + fill[1] + fill[2] + fill[3];
    for (int t = 0; t < 16; ++t) s += acc[t][lane & 3];
    if (lane == 0) { out[(blockIdx.x * WAVES + wave) * 2] = t1 - t0; out[(blockIdx.x * WAVES + wave) * 2 + 1] = (unsigned long long)s; }
}

template <int WAVES, int FINE>
static void run16f(unsigned long long* out, const float* src, int wgs_per_cu, const char* what) {
    const int chunks = 400, grid = 256 * wgs_per_cu;
    auto kern = k16f<WAVES, FINE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) { kern<<<grid, WAVES * 64, 2 * 8192 * 4 + (wgs_per_cu == 2 ? 12 * 1024 : 0), 0>>>(out, chunks, src); hipDeviceSynchronize(); }
    const double cyc = median_cycles(out, grid * WAVES);
    const double per_simd = chunks * 128.0 * 32.0 * (WAVES * wgs_per_cu / 4);
    printf("16x16x4  waves/WG %d  WG/CU %d  (%d waves/SIMD)  SPREAD%d %-40s: %7.0f cycles/wave  pipe busy %.3f  (%.1f cycles per own MFMA)\n", WAVES, wgs_per_cu,
           WAVES * wgs_per_cu / 4, FINE, what, cyc, per_simd / cyc, cyc / (chunks * 128.0));
}


// k16 V4 (lumped form) with two changes, separately switchable: EARLY = the chunk's rendezvous is taken BEFORE the last MFMA group of the chunk
// (whose fragments are in registers), and the next chunk's first fragments are requested right behind it -- the group's eight MFMAs cover the
// LDS round trip that otherwise follows every barrier; BIG = 64-deep chunks (64 KiB, a barrier per 256 MFMAs per wave; the ring then takes
// 128 KiB, so one workgroup per CU).
template <int WAVES, bool EARLY, bool BIG>
__global__ void __launch_bounds__(WAVES * 64, (WAVES * 64) / 256) k16e(unsigned long long* out, int chunks, const float* src) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TILES = BIG ? 4 : 2;                 // input tiles (16 k each) per chunk
    constexpr int CF = TILES * 4096;                   // floats per chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * CF; i += WAVES * 64) lds[i] = src[i & 4095];
    __syncthreads();
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 x = {src[lane], src[64 + lane], src[128 + lane], src[192 + lane]};
    float fill[4] = {1.f, 2.f, 3.f, 4.f};
    const float* wrow = lds + (lane & 15) * 32 + (lane >> 4) * 4;
    int cur = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    f32x4 a0 = *reinterpret_cast<const f32x4*>(wrow), a1 = *reinterpret_cast<const f32x4*>(wrow + 512);
    for (int c = 0; c < chunks / (BIG ? 2 : 1); ++c) {
        const float* wb = wrow + cur * CF;
        if (!EARLY) { a0 = *reinterpret_cast<const f32x4*>(wb); a1 = *reinterpret_cast<const f32x4*>(wb + 512); }
#pragma unroll
        for (int tile = 0; tile < TILES; ++tile) {
            const float* wt = wb + (tile >> 1) * 8192 + (tile & 1) * 16;
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                f32x4 n0 = a0, n1 = a1;
                WAIT_LDS();
                const bool last_group = (tile == TILES - 1 && t == 14);
                if (EARLY && last_group) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    cur ^= 1;
                    const float* wn = wrow + cur * CF;
                    n0 = *reinterpret_cast<const f32x4*>(wn); n1 = *reinterpret_cast<const f32x4*>(wn + 512);
                } else if (t + 2 < 16) {
                    n0 = *reinterpret_cast<const f32x4*>(wt + (t + 2) * 512); n1 = *reinterpret_cast<const f32x4*>(wt + (t + 3) * 512);
                } else if (tile + 1 < TILES) {
                    const float* wn = wb + ((tile + 1) >> 1) * 8192 + ((tile + 1) & 1) * 16;
                    n0 = *reinterpret_cast<const f32x4*>(wn); n1 = *reinterpret_cast<const f32x4*>(wn + 512);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], x[r], acc[t], 0, 0, 0);
                    acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], x[r], acc[t + 1], 0, 0, 0);
                }
                if (t & 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) fill[q] = fill[q] * 1.0001f + 0.5f;
                    fill[0] += fill[3];
                }
                // the DMA pieces of the next chunk: behind the groups of the chunk's first tile(s) -- with EARLY they target the buffer every wave
                // left at the rendezvous inside the PREVIOUS chunk's last group
                const int gidx = tile * 8 + t / 2;                                   // group index in the chunk
                const int npieces = CF * 4 / (WAVES * 1024);
                const int pstep = (WAVES == 8) ? 2 : 1;
                if (gidx % pstep == 0 && gidx / pstep < npieces && !(EARLY && last_group)) {
                    typedef const __attribute__((address_space(1))) void* gptr_t;
                    typedef __attribute__((address_space(3))) void* lptr_t;
                    const int i = gidx / pstep;
                    const int wv = __builtin_amdgcn_readfirstlane(wave);
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + ((size_t)(c & 7) * CF + (i * WAVES * 64 + tid) * 4) % (1 << 20)),
                                                     (lptr_t)(lds + (cur ^ 1) * CF + (i * WAVES * 64 + wv * 64) * 4), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                a0 = n0; a1 = n1;
            }
        }
        if (!EARLY) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); cur ^= 1; }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = fill[0] + fill[1] + fill[2] + fill[3];
    for (int t = 0; t < 16; ++t) s += acc[t][lane & 3];
    if (lane == 0) { out[(blockIdx.x * WAVES + wave) * 2] = t1 - t0; out[(blockIdx.x * WAVES + wave) * 2 + 1] = (unsigned long long)s; }
}

template <int WAVES, bool EARLY, bool BIG>
static void run16e(unsigned long long* out, const float* src, int wgs_per_cu, const char* what) {
    const int chunks = 400, grid = 256 * wgs_per_cu;
    auto kern = k16e<WAVES, EARLY, BIG>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t ldsb = (BIG ? 4 : 2) * 8192 * 4 + ((wgs_per_cu == 2 && !BIG) ? 12 * 1024 : 0);
    for (int rep = 0; rep < 2; ++rep) { kern<<<grid, WAVES * 64, ldsb, 0>>>(out, chunks, src); hipDeviceSynchronize(); }
    const double cyc = median_cycles(out, grid * WAVES);
    const double per_simd = chunks * 128.0 * 32.0 * (WAVES * wgs_per_cu / 4);
    printf("16x16x4  waves/WG %d  WG/CU %d  (%d waves/SIMD)  early-barrier %d  64-deep chunks %d  %-26s: %7.0f cycles/wave  pipe busy %.3f\n", WAVES, wgs_per_cu,
           WAVES * wgs_per_cu / 4, (int)EARLY, (int)BIG, what, cyc, per_simd / cyc);
}

static double median_cycles(unsigned long long* dout, int n) {
    std::vector<unsigned long long> h(2 * n);
    hipMemcpy(h.data(), dout, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> c(n);
    for (int i = 0; i < n; ++i) c[i] = h[2 * i];
    std::sort(c.begin(), c.end());
    return (double)c[n / 2];
}

template <int WAVES, int V>
static void run16(unsigned long long* out, const float* src, int wgs_per_cu, const char* what) {
    const int chunks = 400, grid = 256 * wgs_per_cu;
    auto kern = k16<WAVES, V>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) { kern<<<grid, WAVES * 64, 2 * 8192 * 4 + (wgs_per_cu == 2 ? 12 * 1024 : 0), 0>>>(out, chunks, src); hipDeviceSynchronize(); }
    const double cyc = median_cycles(out, grid * WAVES);
    // slots of SIMD time: a wave issues chunks x 128 MFMAs of 32 cycles; waves per SIMD = WAVES x wgs_per_cu / 4
    const double per_simd = chunks * 128.0 * 32.0 * (WAVES * wgs_per_cu / 4);
    printf("16x16x4  waves/WG %d  WG/CU %d  (%d waves/SIMD)  V%d %-44s: %7.0f cycles/wave  pipe busy %.3f  (%.1f cycles per own MFMA)\n", WAVES, wgs_per_cu,
           WAVES * wgs_per_cu / 4, V, what, cyc, per_simd / cyc, cyc / (chunks * 128.0));
}
template <int V>
static void run32(unsigned long long* out, const float* src, const char* what) {
    const int chunks = 400, grid = 256;
    auto kern = k32<V>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) { kern<<<grid, 256, 2 * 8192 * 4 + 80 * 1024, 0>>>(out, chunks, src); hipDeviceSynchronize(); }
    const double cyc = median_cycles(out, grid * 4);
    printf("32x32x2  one wave/SIMD                      V%d %-44s: %7.0f cycles/wave  pipe busy %.3f  (%.1f cycles per MFMA)\n", V, what, cyc,
           chunks * 128.0 * 64.0 / cyc, cyc / (chunks * 128.0));
}

int main() {
    unsigned long long* out; float* src;
    hipMalloc(&out, 512 * 8 * 2 * 8); hipMalloc(&src, (1 << 20) * 4 + 65536);
    std::vector<float> h((1 << 20) + 16384);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;       // random-ish operands (the clock depends on the data)
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run32<0>(out, src, "bare MFMAs");
    run32<1>(out, src, "+ LDS fragments");
    run32<3>(out, src, "+ barrier per chunk");
    run32<4>(out, src, "+ LDS-DMA ring, vmcnt(0) + barrier");
    run16<4, 0>(out, src, 1, "bare MFMAs");
    run16<8, 0>(out, src, 1, "bare MFMAs");
    run16<4, 1>(out, src, 1, "+ LDS fragments");
    run16<8, 1>(out, src, 1, "+ LDS fragments");
    run16<8, 2>(out, src, 1, "+ epilogue VALU slices");
    run16<8, 3>(out, src, 1, "+ barrier per chunk");
    run16<8, 4>(out, src, 1, "+ LDS-DMA ring, vmcnt(0) + barrier");
    run16<8, 5>(out, src, 1, "+ stagger: waves 4-7 sleep after barrier");
    run16<4, 4>(out, src, 1, "+ LDS-DMA ring, vmcnt(0) + barrier");
    run16<4, 4>(out, src, 2, "two independent workgroups per CU");
    run16f<8, 2>(out, src, 1, "fragments + VALU, spread (cf. V2)");
    run16f<4, 2>(out, src, 1, "fragments + VALU, spread, ONE wave/SIMD");
    run16f<8, 1>(out, src, 1, "+ DMA ring + barrier, spread (cf. V4)");
    run16f<4, 1>(out, src, 1, "+ DMA ring + barrier, spread, one wave/SIMD");
    run16f<4, 1>(out, src, 2, "the same, two workgroups per CU");
    run16e<8, false, false>(out, src, 1, "(= V4, carried fragments)");
    run16e<8, true, false>(out, src, 1, "");
    run16e<8, false, true>(out, src, 1, "");
    run16e<8, true, true>(out, src, 1, "");
    run16e<4, true, false>(out, src, 2, "");
    run16e<4, true, false>(out, src, 1, "one wave per SIMD");
    return 0;
}
