// Shared pieces of the two-waves-per-SIMD fp32 kernels (forward: snr_mlp16.hip, backward: snr_mlp16_bwd.hip): the LDS map, the scalar-base
// LDS-DMA of the weight ring, and the v_mfma_f32_16x16x4_f32 tile product with its fragment pipeline.
#pragma once
#include <stdlib.h>
#include "snr_mlp_core.hpp"

namespace snr {

constexpr int PE_WAVE16 = 16 * PE_ROW;               // per-wave positional-encoding scratch: 16 points
constexpr int WBUF16 = 256 * KC;                     // floats per ring buffer: the forward stream's largest chunk, 256 rows x 32 (32 KiB)

// LDS map (float offsets; the kernel declares no static LDS, the host computes the map per launch and passes the byte size as the
// launch's dynamic LDS).  WAVES = 8 (one 512-thread workgroup per CU): [ring 0 | ring 1 | scratch | composite | bias + heads | latent | zero].
// WAVES = 4 (TWO 256-thread workgroups per CU, 80 KiB each at most): the scratch (needed until the operand registers are read) lies over
// ring buffer 1 (first written by the DMA of chunk 1, behind a barrier) and the composite scratch (needed after the last chunk) over ring
// buffer 0; the bias and latent blocks are sized by the decoder's own layer count.
struct Lds16 { int ring1, scratch, comp, bias, lat, zero, total; };
inline Lds16 make_lds16(int waves, int n_mfma_layers, int n_lat) {
    Lds16 o;
    o.ring1 = WBUF16;
    int p = 2 * WBUF16;
    if (waves == 8) { o.scratch = p; p += 8 * PE_WAVE16; o.comp = p; p += 128 * COMP_STRIDE; }
    else { o.scratch = o.ring1; o.comp = 0; }
    p = (p + 3) & ~3;
    o.bias = p; p += (n_mfma_layers + 3) * 256;
    o.lat = p; p += (n_lat <= LDS_LAT_ROWS ? n_lat : 0) * 256;
    o.zero = p; p += 256;
    o.total = p;
    return o;
}

// LDS-DMA of the weight ring through inline asm with a SCALAR base (global_load_lds_dwordx4 voff, s[base:base+1] offset:imm; M0 = the LDS
// address): wave w copies the contiguous slice w of a chunk (rows x 128 B / WAVES bytes: 4 or 8 pieces of 1 KiB for 256 rows), the
// per-lane VGPR offset (lane x 16 + 4096) is made once per kernel and the piece is selected by the immediate, which moves the global and
// the LDS address together (-4096 .. +3072).  The builtin form computes a 64-bit per-lane address on the VALU for every piece -- and on
// this chip VALU instructions compete with the fp32 MFMAs for the ALUs.  EVERY LDS-DMA of the kernel takes this form: the compiler must
// never have a use of M0 of its own (it would assume M0 survives an asm statement).  Completion is counted by hand (ring_turn: vmcnt(0)).
template <int K>
__device__ __forceinline__ void dma_piece_imm(unsigned voff, const void* sbase, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" :: "v"(voff), "s"(sbase), "s"(m0v), "n"(1024 * K - 4096) : "memory");
}
__device__ __forceinline__ void dma_piece(int k, unsigned voff, const void* sbase, unsigned m0v) {      // k is a constant after unrolling
    switch (k) {
        case 0: dma_piece_imm<0>(voff, sbase, m0v); break;
        case 1: dma_piece_imm<1>(voff, sbase, m0v); break;
        case 2: dma_piece_imm<2>(voff, sbase, m0v); break;
        case 3: dma_piece_imm<3>(voff, sbase, m0v); break;
        case 4: dma_piece_imm<4>(voff, sbase, m0v); break;
        case 5: dma_piece_imm<5>(voff, sbase, m0v); break;
        case 6: dma_piece_imm<6>(voff, sbase, m0v); break;
        default: dma_piece_imm<7>(voff, sbase, m0v); break;
    }
}
struct Dma16 {
    unsigned voff;       // lane * 16 + 4096
    unsigned lds0;       // LDS byte address of lds[0] (uniform)
    int wave;            // uniform
};
// piece i of this wave's slice of a chunk of `rows` rows: bytes [wave * SL + i KiB, + 1 KiB), SL = rows * 128 / WAVES
template <int WAVES>
__device__ __forceinline__ void chunk_piece16(const Dma16& d, const float* __restrict__ g, const float* lds_dst, const float* lds_base, int rows, int i) {
    const unsigned sl = (unsigned)(rows * 128 / WAVES) * (unsigned)d.wave;
    const char* sbase = reinterpret_cast<const char*>(g) + sl;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(d.lds0 + (unsigned)((lds_dst - lds_base) * 4) + sl + 4096u);
    dma_piece(i, d.voff, sbase, m0v);
}
// a whole chunk of `rows` x 128 B at once (prologue and enc_xyz; inside the 256-wide layers the pieces go out between MFMA groups)
template <int WAVES>
__device__ __forceinline__ void chunk_dma16(const Dma16& d, const float* __restrict__ g, const float* lds_dst, const float* lds_base, int rows) {
#pragma unroll
    for (int i = 0; i < 32 / WAVES; ++i)
        if (i * 8 * WAVES < rows) chunk_piece16<WAVES>(d, g, lds_dst, lds_base, rows, i);
}
// one 1 KiB row (biases, heads, latent rows) by ONE wave
__device__ __forceinline__ void row_dma16(const Dma16& d, const float* __restrict__ g, const float* lds_dst, const float* lds_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(d.lds0 + (unsigned)((lds_dst - lds_base) * 4) + 4096u);
    dma_piece_imm<0>(d.voff, g, m0v);
}

// (ray, sample, object) of point pl (0 .. wgp - 1) of workgroup tile_wg without per-lane 64-bit divisions (~150 VALU instructions each, and on
// this chip every VALU instruction comes out of the fp32 MFMAs' time): in the fused render S is a power of two that divides the workgroup's
// points, so the ray is a shift away from the workgroup's first ray, and the object follows from ONE uniform division (scalar ALU) per workgroup.
// Lanes past the end of the launch hold its last point, like everywhere else.
struct PointId { long long ray, obj; int s; };
__device__ __forceinline__ PointId point_id(const RayGeom& g, long long tile_wg, int wgp, int pl, bool live) {
    const int ls = 31 - __builtin_clz((unsigned)g.S);
    const long long ray0 = tile_wg * (long long)(wgp >> ls);
    const long long obj0 = ray0 / g.rays_per_obj;
    const long long rem0 = ray0 - obj0 * g.rays_per_obj;
    const int dr = pl >> ls;
    const long long r = rem0 + dr;
    PointId p;
    p.ray = ray0 + dr;
    p.s = pl & (g.S - 1);
    p.obj = obj0 + (g.rays_per_obj >= wgp ? (long long)(r >= g.rays_per_obj) : (long long)((unsigned)r / (unsigned)g.rays_per_obj));
    if (!live) { p.ray = g.n_rays - 1; p.s = g.S - 1; p.obj = (g.n_rays - 1) / g.rays_per_obj; }
    return p;
}

struct Ring16 {
    int cur;             // LDS buffer holding the chunk about to be consumed
    int aoff[2];         // this lane's float offset of the 16-byte slot (4 dT + kg) of row m in a chunk, swizzle applied (dT = 0, 1)
};

#define SNR16_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xc07f)       /* s_waitcnt lgkmcnt(0), as an instruction the compiler's own wait insertion sees */

// 64 MFMAs of one input tile: accC[t] += W[16 t .. 16 t + 15][k-slices of the tile] * x, tiles taken in pairs so that two MFMAs on the same
// accumulator are two issues apart (dependent latency 40 cycles > the 32-cycle issue); LAST: the finished sums go to accD (the dead previous set).
// Fragment pipeline: (a0, a1) = the tile's first pair, REQUESTED by the caller; every pair is waited for at the top of its group -- at that
// point only that pair is outstanding, requested a whole group (>= 256 cycles) earlier -- and the next pair (the next TILE's first pair from
// `wnext` at the end) is requested before the group's eight MFMAs.  Left to the compiler, the request follows the MFMAs and the wait
// (lgkmcnt(0), not a counted one) sits right behind the request: an LDS round trip exposed per group in both waves of the SIMD at once.
struct NoBetween { __device__ __forceinline__ void operator()(int) const {} };
// FLAGS: bit 0 (LAST): the finished sums go to accD; bit 1 (ZERO): the sums start from zero -- every tile's first MFMA takes the constant 0 as
// its C operand, no zeroed register set.
constexpr int TM_LAST = 1, TM_ZERO = 2;
template <int NT, int FLAGS, class Between = NoBetween, int NA = 16, int ND = 16>
__device__ __forceinline__ void tile_mma(f32x4 (&accC)[NA], f32x4 (&accD)[ND], const f32x4& x, const float* wrow /* chunk + aoff[dT] */,
                                         f32x4& a0, f32x4& a1, const float* wnext, Between&& between = NoBetween()) {
    constexpr bool LAST = FLAGS & TM_LAST, ZERO = FLAGS & TM_ZERO;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
        SNR16_WAIT_LDS();
        f32x4 n0 = a0, n1 = a1;
        if (t + 2 < NT) {
            n0 = *reinterpret_cast<const f32x4*>(wrow + (t + 2) * 16 * KC);
            n1 = *reinterpret_cast<const f32x4*>(wrow + (t + 3) * 16 * KC);
        } else if (wnext) {
            n0 = *reinterpret_cast<const f32x4*>(wnext);
            n1 = *reinterpret_cast<const f32x4*>(wnext + 16 * KC);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (LAST && r == 3) {
                accD[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], x[r], accC[t], 0, 0, 0);
                accD[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], x[r], accC[t + 1], 0, 0, 0);
            } else {
                accC[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], x[r], (ZERO && r == 0) ? zero4 : accC[t], 0, 0, 0);
                accC[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], x[r], (ZERO && r == 0) ? zero4 : accC[t + 1], 0, 0, 0);
            }
        }
        // the group's slice of non-matrix work, in program order BEHIND its eight MFMAs: it issues while they -- and the partner wave's --
        // occupy the matrix pipe (an in-order wave reaches its next MFMA >= 64 cycles later with two waves per SIMD)
        between(t / 2);
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0; a1 = n1;
    }
}

__device__ __forceinline__ void first_pair(f32x4& a0, f32x4& a1, const float* wrow) {
    a0 = *reinterpret_cast<const f32x4*>(wrow);
    a1 = *reinterpret_cast<const f32x4*>(wrow + 16 * KC);
}

__device__ __forceinline__ void ring_turn(Ring16& p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    p.cur ^= 1;
}


}  // namespace snr
