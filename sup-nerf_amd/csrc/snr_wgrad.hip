// Weight gradients of the per-point decoder layers for gfx950 (training mode, SURVEY 8 a9 mode B / f2):
//     dW[i][j] = sum_p G[p][i] * X[p][j],      db[i] = sum_p G[p][i]
// with G (P, n_out) the gradient wrt a layer's pre-activation and X (P, n_in) the layer's input, both row-major fp32 as the training
// forward / backward kernels wrote them.  This is the weight half of loss_total.mean().backward() (src/trainer_unified_nuscenes.py:334),
// replacing one library GEMM + one column-sum per layer.
//
// Split-K over the points, exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak): a workgroup of 4 waves owns one
// slice of points and the whole (<= 256 x 256) product, wave (wr, wc) the 128 x 128 quadrant = 16 accumulator tiles (256 AGPRs).
// The reduction dimension is the POINT index, so the row-major operands need no transpose: lane (n, h) takes G[p + h][4n .. 4n+3] and
// X[p + h][4n .. 4n+3] with one 16-byte load each and these ARE the A / B operands of a k = 2 step for the four row tiles
// {4n + e} x four column tiles {4m + e'} (the tiles interleave rows / columns with stride 4; the final store undoes it with 16-byte
// stores).  Per k-step and wave: 2 loads, 16 MFMAs (1024 cycles) -- matrix-bound; G and X are read from HBM once.
// The per-slice partial products go to a workspace and are summed over the slices in slice order (deterministic, no atomics).
#include "snr_device.hpp"
#include "snr_host.hpp"

namespace snr {

constexpr int WG_DEPTH = 8;            // k-steps per operand set: one set in flight per wave while the other is consumed (4, 12 and 16 measured the same)

// Operand addressing shared by both arithmetics: a wave-uniform base (SGPRs, advanced per k-step) + a loop-invariant 32-bit lane offset,
// so a load is one instruction with no address arithmetic.  Loads are unconditional and unmasked in the main loop: lanes whose rows /
// columns lie outside the layer (n_out < 256, n_in < 256) read in-bounds neighbours and fill partial rows / columns nobody reads, and the
// prefetch that runs past the slice is pointed back at its start and discarded.  Only the last, ragged k-step of a slice masks.
struct WgLane {
    unsigned g_off, x_off;      // bytes: (lane's point row within a k-step) * ld + lane's 4 columns
};

__global__ void __launch_bounds__(256, 1)
wgrad_mfma_kernel(const float* __restrict__ G, long long ldg, int n_out, const float* __restrict__ X, long long ldx, int n_in,
                  long long n_points, long long points_per_slice, float* __restrict__ part_w /* [slice][256][256] */,
                  float* __restrict__ part_b /* [slice][256] or null */) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    if (128 * wr >= n_out || 128 * wc >= n_in) return;                 // this wave's quadrant lies outside the layer (wave-uniform)
    const long long p_begin = blockIdx.x * points_per_slice;
    const long long p_end = p_begin + points_per_slice < n_points ? p_begin + points_per_slice : n_points;
    const int row0 = 128 * wr + 4 * n, col0 = 128 * wc + 4 * n;
    const int row_c = row0 < n_out ? row0 : 0, col_c = col0 < n_in ? col0 : 0;
    const unsigned g_off = (unsigned)((h * ldg + row_c) * 4), x_off = (unsigned)((h * ldx + col_c) * 4);
    const char* Gb = reinterpret_cast<const char*>(G);
    const char* Xb = reinterpret_cast<const char*>(X);
    f32x16 acc[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    constexpr int SET = 2 * WG_DEPTH;                                  // points per operand set
    constexpr int TRIP = 2 * SET;                                      // points per trip of the main loop: two sets, each consumed in place
    const long long full_end = p_begin + (p_end - p_begin) / TRIP * TRIP;
    auto load = [&](long long p, f32x4& a, f32x4& b) {                 // k-step at points p, p + 1 (this lane: p + h); p is wave-uniform
        const long long pc = p + 2 <= p_end ? p : p_begin;             // past the slice: any valid k-step, the values are dropped
        a = *reinterpret_cast<const f32x4*>(Gb + pc * ldg * 4 + g_off);
        b = *reinterpret_cast<const f32x4*>(Xb + pc * ldx * 4 + x_off);
    };
    auto mma = [&](const f32x4& av, const f32x4& bv) {
        bsum += av;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[f], acc[e][f], 0, 0, 0);
    };
    // Two operand sets of WG_DEPTH k-steps each: while one set is consumed IN PLACE the other one lands (WG_DEPTH x 1024 MFMA cycles cover
    // an HBM round trip), and a set's registers are requested again right after the MFMAs that read them.  (A single set refilled slot by
    // slot behind a register copy made LLVM rotate the whole set through v_mov at the end of every trip, behind an s_waitcnt vmcnt(0):
    // every trip then waited for the loads it had just issued; this form: 0.570 -> 0.544 ms per 256 x 256 layer at 524 288 points.  The rest
    // of the distance to the 0.44 ms of matrix time is the clock: the counters put the matrix pipe at 0.85 busy at ~2.08 GHz.)
    f32x4 a0[WG_DEPTH], b0[WG_DEPTH], a1[WG_DEPTH], b1[WG_DEPTH];
    if (p_begin < full_end) {
#pragma unroll
        for (int d = 0; d < WG_DEPTH; ++d) load(p_begin + 2 * d, a0[d], b0[d]);
    }
    for (long long p = p_begin; p < full_end; p += TRIP) {
#pragma unroll
        for (int d = 0; d < WG_DEPTH; ++d) {
            load(p + SET + 2 * d, a1[d], b1[d]);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0[d], b0[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long long pn = p + TRIP < full_end ? p + TRIP : p_begin;          // past the last trip: a valid set, dropped
#pragma unroll
        for (int d = 0; d < WG_DEPTH; ++d) {
            load(pn + 2 * d, a0[d], b0[d]);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1[d], b1[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (long long p = full_end; p < p_end; p += 2) {                  // ragged tail: fewer than TRIP points, masked per lane
        const long long q = p + h;
        const long long qc = q < p_end ? q : p_end - 1;
        const float m = q < p_end ? 1.f : 0.f;
        const f32x4 av = *reinterpret_cast<const f32x4*>(Gb + qc * ldg * 4 + (unsigned)(row_c * 4)) * m;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(Xb + qc * ldx * 4 + (unsigned)(col_c * 4)) * m;
        mma(av, bv);
    }
    // tile (e, f): D[r][c] = sum_p G[p][128 wr + 4 r + e] X[p][128 wc + 4 c + f];  this lane: c = n, r = (reg & 3) + 8 (reg >> 2) + 4 h
    float* out = part_w + (long long)blockIdx.x * 256 * 256;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const f32x4 v = {acc[e][0][reg], acc[e][1][reg], acc[e][2][reg], acc[e][3][reg]};
            *reinterpret_cast<f32x4*>(out + (128 * wr + 4 * r + e) * 256 + 128 * wc + 4 * n) = v;
        }
    if (part_b && wc == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[e] = sum_halves(bsum[e]);
        if (h == 0) *reinterpret_cast<f32x4*>(part_b + (long long)blockIdx.x * 256 + row0) = bsum;
    }
}

// The same product in split-bf16 arithmetic (SNR_BF16X3): every fp32 operand as bf16 hi + lo, three v_mfma_f32_32x32x16_bf16 per
// product, fp32 accumulate (relative operand error ~2^-17, like the render kernels' fast path).  A k-step is 16 points: lane (n, h)
// loads G[p + 8h + j][4n .. 4n+3] and X[p + 8h + j][4n .. 4n+3] for j = 0..7 (sixteen 16-byte loads) and holds, for each of the four
// interleaved row tiles {4n + e}, exactly the A fragment of the 32x32x16 MFMA (row 4n + e, k = 8h + j) -- again no transpose.
// 48 MFMAs of 32 cycles per 16 points instead of 128 of 64: 5.3x less matrix time, which leaves the kernel bound by reading G and X
// from HBM once (2 KB per point and layer).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8(const f32x4 (&v)[8], int e, bf16x8_t& hi, bf16x8_t& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[j][e];
        const __bf16 b = (__bf16)x;
        hi[j] = b;
        lo[j] = (__bf16)(x - (float)b);
    }
}

__global__ void __launch_bounds__(256, 1)
wgrad_bf16x3_kernel(const float* __restrict__ G, long long ldg, int n_out, const float* __restrict__ X, long long ldx, int n_in,
                    long long n_points, long long points_per_slice, float* __restrict__ part_w, float* __restrict__ part_b) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    if (128 * wr >= n_out || 128 * wc >= n_in) return;
    const long long p_begin = blockIdx.x * points_per_slice;
    const long long p_end = p_begin + points_per_slice < n_points ? p_begin + points_per_slice : n_points;
    const int row0 = 128 * wr + 4 * n, col0 = 128 * wc + 4 * n;
    const int row_c = row0 < n_out ? row0 : 0, col_c = col0 < n_in ? col0 : 0;
    unsigned g_off[8], x_off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { g_off[j] = (unsigned)(((8 * h + j) * ldg + row_c) * 4); x_off[j] = (unsigned)(((8 * h + j) * ldx + col_c) * 4); }
    const char* Gb = reinterpret_cast<const char*>(G);
    const char* Xb = reinterpret_cast<const char*>(X);
    f32x16 acc[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const long long full_end = p_begin + (p_end - p_begin) / 32 * 32;
    auto load = [&](long long p, f32x4 (&a)[8], f32x4 (&b)[8]) {        // the 16 points p .. p+15 (p wave-uniform); this lane: p + 8h + j
        const char* gb = Gb + p * ldg * 4;
        const char* xb = Xb + p * ldx * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = *reinterpret_cast<const f32x4*>(gb + g_off[j]);
            b[j] = *reinterpret_cast<const f32x4*>(xb + x_off[j]);
        }
    };
    auto compute = [&](const f32x4 (&a)[8], const f32x4 (&b)[8]) {
        bf16x8_t bh[4], bl[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) split8(b, f, bh[f], bl[f]);
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += a[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x8_t ah, al;
            split8(a, e, ah, al);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[f], acc[e][f], 0, 0, 0);
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[f], acc[e][f], 0, 0, 0);
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[f], acc[e][f], 0, 0, 0);
            }
        }
    };
    f32x4 a0[8], b0[8], a1[8], b1[8];
    if (p_begin < full_end) load(p_begin, a0, b0);
    for (long long p = p_begin; p < full_end; p += 32) {        // two k-steps per trip: one set of operands in flight while the other is used
        // (sched_barrier: LLVM otherwise hoists the conversion of the set just requested above the MFMAs of the current one and waits
        //  for the loads it has only just issued)
        load(p + 16, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        compute(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load(p + 32 < full_end ? p + 32 : p_begin, a0, b0);     // past the last trip: a valid k-step, dropped
        __builtin_amdgcn_sched_barrier(0);
        compute(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (long long p = full_end; p < p_end; p += 16) {          // ragged tail (< 32 points): masked per lane
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long q = p + 8 * h + j;
            const long long qc = q < p_end ? q : p_end - 1;
            const float m = q < p_end ? 1.f : 0.f;
            a0[j] = *reinterpret_cast<const f32x4*>(Gb + qc * ldg * 4 + (unsigned)(row_c * 4)) * m;
            b0[j] = *reinterpret_cast<const f32x4*>(Xb + qc * ldx * 4 + (unsigned)(col_c * 4)) * m;
        }
        compute(a0, b0);
    }
    float* out = part_w + (long long)blockIdx.x * 256 * 256;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const f32x4 v = {acc[e][0][reg], acc[e][1][reg], acc[e][2][reg], acc[e][3][reg]};
            *reinterpret_cast<f32x4*>(out + (128 * wr + 4 * r + e) * 256 + 128 * wc + 4 * n) = v;
        }
    if (part_b && wc == 0) {           // (the bias gradient stays an exact fp32 sum)
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[e] = sum_halves(bsum[e]);
        if (h == 0) *reinterpret_cast<f32x4*>(part_b + (long long)blockIdx.x * 256 + row0) = bsum;
    }
}

// sum of the per-slice partials -> dW (n_out x n_in, leading dimension ld_dw) and db.  Block = 256 outputs x 4 slice groups (1024 threads):
// group g adds the slices k = g (mod 4) with eight loads in flight, the groups meet in LDS; the association is fixed, so the result is the
// same bits every run.  Blocks 0..255 cover the 256 x 256 outputs, block 256 the bias.
__global__ void __launch_bounds__(1024) wgrad_reduce_kernel(const float* __restrict__ part_w, const float* __restrict__ part_b, int n_slices, int n_out,
                                                            int n_in, float* __restrict__ dW, long long ld_dw, float* __restrict__ db) {
    __shared__ float red[4][256];
    const int j = threadIdx.x & 255, g = threadIdx.x >> 8;
    const bool bias = blockIdx.x == 256;
    if (bias && !(db && part_b)) return;
    const int i = blockIdx.x;                                   // output row (weights) -- unused for the bias block
    const float* src = bias ? part_b : part_w + i * 256;
    const long long stride = bias ? 256 : 65536;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias || (i < n_out && j < n_in)) {
        int k = g;
        for (; k + 28 < n_slices; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += src[(long long)(k + 4 * u) * stride + j];
        }
        for (int u = 0; k < n_slices; k += 4, ++u) s[u & 7] += src[(long long)k * stride + j];
    }
    red[g][j] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (g != 0) return;
    const float v = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    if (bias) { if (j < n_out) db[j] = v; }
    else if (i < n_out && j < n_in) dW[i * ld_dw + j] = v;
}

// narrow outputs (the density head: 1 row, the colour head: 3 rows): one thread per input column, points strided over the slices;
// X is streamed once with coalesced loads, the <= 4 gradient values of a point are wave-uniform (scalar) loads.  Whole groups of eight
// points run without a single condition (NO = n_out is a template parameter, a column past n_in reads column 0 and its sums are never
// looked at): the conditional form spent the loop on ~80 scalar branches per group (282 us for the two heads at 524 288 points,
// 215 us now).
template <int NO>
__global__ void __launch_bounds__(256) wgrad_small_kernel(const float* __restrict__ G, long long ldg, const float* __restrict__ X, long long ldx,
                                                          int n_in, long long n_points, long long points_per_slice, float* __restrict__ part_w,
                                                          float* __restrict__ part_b) {
    const int j = threadIdx.x;
    const int jc = j < n_in ? j : 0;
    const long long p_begin = blockIdx.x * points_per_slice;
    const long long p_end = p_begin + points_per_slice < n_points ? p_begin + points_per_slice : n_points;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
    long long p = p_begin;
    constexpr int WD = 8;                                      // (4, 16, 32 measured the same once the conditions were gone)
    for (; p + WD <= p_end; p += WD) {                         // WD points' loads in flight
        float x[WD], g[WD][NO];
#pragma unroll
        for (int u = 0; u < WD; ++u) x[u] = X[(p + u) * ldx + jc];
#pragma unroll
        for (int u = 0; u < WD; ++u)
#pragma unroll
            for (int i = 0; i < NO; ++i) g[u][i] = G[(p + u) * ldg + i];
#pragma unroll
        for (int u = 0; u < WD; ++u)
#pragma unroll
            for (int i = 0; i < NO; ++i) { acc[i] += g[u][i] * x[u]; bs[i] += g[u][i]; }
    }
    for (; p < p_end; ++p) {                                   // ragged tail (< WD points)
        const float xv = X[p * ldx + jc];
#pragma unroll
        for (int i = 0; i < NO; ++i) { const float gv = G[p * ldg + i]; acc[i] += gv * xv; bs[i] += gv; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) part_w[((long long)blockIdx.x * 4 + i) * 256 + j] = acc[i];
    if (j == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) part_b[(long long)blockIdx.x * 4 + i] = bs[i];
    }
}
// grid = n_out, 1024 threads: column j = tid & 255, slice group g = tid >> 8 sums the slices k = g (mod 4) with eight loads in flight,
// the four groups meet in LDS in a fixed order (deterministic)
__global__ void __launch_bounds__(1024) wgrad_small_reduce_kernel(const float* __restrict__ part_w, const float* __restrict__ part_b, int n_slices, int n_out,
                                                                  int n_in, float* __restrict__ dW, long long ld_dw, float* __restrict__ db) {
    __shared__ float red[4][256];
    const int j = threadIdx.x & 255, g = threadIdx.x >> 8, i = blockIdx.x;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = g;
    for (; k + 28 < n_slices; k += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += part_w[((long long)(k + 4 * u) * 4 + i) * 256 + j];
    }
    for (int u = 0; k < n_slices; k += 4, ++u) s[u & 7] += part_w[((long long)k * 4 + i) * 256 + j];
    red[g][j] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (g == 0 && j < n_in) dW[i * ld_dw + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    if (db) {           // the bias: every thread takes the slices q = tid (mod 1024), then a fixed-order sum of the 1024 partials
        __syncthreads();
        float t = 0.f;
        for (int q = threadIdx.x; q < n_slices; q += 1024) t += part_b[(long long)q * 4 + i];
        red[g][j] = t;
        __syncthreads();
        if (threadIdx.x < 64) {
            float u = 0.f;
            for (int q = threadIdx.x; q < 1024; q += 64) u += red[q >> 8][q & 255];
            u = wave_sum(u);
            if (threadIdx.x == 0) db[i] = u;
        }
    }
}

}  // namespace snr

using namespace snr;

static void wgrad_plan(int64_t n_points, int n_out, long long* pps, int* n_slices) {
    // one slice per compute unit and pass where the points allow it; a slice is a whole number of k-step groups
    const long long unit = 32;      // whole main-loop trips of both arithmetics (fp32: 2 sets x WG_DEPTH k-steps x 2 points, split-bf16: 2 x 16)
    long long per = (n_points + (n_out >= 32 ? 255 : 1023)) / (n_out >= 32 ? 256 : 1024);      // narrow heads: memory-bound, 4 blocks per CU
    const long long min_per = 512;
    if (per < min_per) per = min_per;
    per = (per + unit - 1) / unit * unit;
    *pps = per;
    *n_slices = (int)((n_points + per - 1) / per);
}

extern "C" {

size_t snr_weight_grad_ws_bytes(int64_t n_points, int n_out, int n_in) {
    (void)n_in;
    if (n_points <= 0 || n_out <= 0) return 256;
    long long pps; int ns;
    wgrad_plan(n_points, n_out, &pps, &ns);
    return (size_t)ns * (n_out >= 32 ? (256 * 256 + 256) : (4 * 256 + 4)) * sizeof(float) + 256;
}

int snr_weight_grad(const float* G, int64_t ldg, int n_out, const float* X, int64_t ldx, int n_in, int64_t n_points,
                    float* dW, int64_t ld_dw, float* db, int precision, void* workspace, size_t ws_bytes, void* stream_) {
    if (precision != SNR_FP32 && precision != SNR_BF16X3) return SNR_E_ARG;
    if (!G || !X || !dW || n_out < 1 || n_in < 1 || n_out > 256 || n_in > 256 || n_points < 0 || ldg < n_out || ldx < n_in || ld_dw < n_in) return SNR_E_ARG;
    hipStream_t st = (hipStream_t)stream_;
    if (!workspace || ws_bytes < snr_weight_grad_ws_bytes(n_points, n_out, n_in)) return SNR_E_WORKSPACE;
    long long pps; int ns;
    wgrad_plan(n_points > 0 ? n_points : 1, n_out, &pps, &ns);
    float* ws = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (n_out >= 32) {
        // 16-byte operand loads: whole groups of 4 rows / columns and aligned rows
        if ((n_out & 3) || (n_in & 3) || (ldg & 3) || (ldx & 3) || ((uintptr_t)G & 15) || ((uintptr_t)X & 15)) return SNR_E_UNSUPPORTED;
        float* part_w = ws; float* part_b = ws + (long long)ns * 65536;
        if (n_points == 0) ns = 0;
        if (ns && precision == SNR_BF16X3) wgrad_bf16x3_kernel<<<ns, 256, 0, st>>>(G, ldg, n_out, X, ldx, n_in, n_points, pps, part_w, db ? part_b : nullptr);
        else if (ns) wgrad_mfma_kernel<<<ns, 256, 0, st>>>(G, ldg, n_out, X, ldx, n_in, n_points, pps, part_w, db ? part_b : nullptr);
        wgrad_reduce_kernel<<<257, 1024, 0, st>>>(part_w, db ? part_b : nullptr, ns, n_out, n_in, dW, ld_dw, db);
    } else {
        if (n_out > 4) return SNR_E_UNSUPPORTED;        /* 5..31 output rows: no such layer in the decoder */
        float* part_w = ws; float* part_b = ws + (long long)ns * 4 * 256;
        if (n_points == 0) ns = 0;
        if (ns && n_out == 1) wgrad_small_kernel<1><<<ns, 256, 0, st>>>(G, ldg, X, ldx, n_in, n_points, pps, part_w, part_b);
        else if (ns && n_out == 2) wgrad_small_kernel<2><<<ns, 256, 0, st>>>(G, ldg, X, ldx, n_in, n_points, pps, part_w, part_b);
        else if (ns && n_out == 3) wgrad_small_kernel<3><<<ns, 256, 0, st>>>(G, ldg, X, ldx, n_in, n_points, pps, part_w, part_b);
        else if (ns) wgrad_small_kernel<4><<<ns, 256, 0, st>>>(G, ldg, X, ldx, n_in, n_points, pps, part_w, part_b);
        wgrad_small_reduce_kernel<<<n_out, 1024, 0, st>>>(part_w, part_b, ns, n_out, n_in, dW, ld_dw, db);
    }
    return snr_check_launch_();
}

}  // extern "C"
