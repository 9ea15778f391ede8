// Split-bf16 ("bf16x3") variant of the fused decoder kernels for gfx950.
//
// Same mathematics and the same register-resident layer chain as the fp32 kernels (snr_mlp*.hip), but every fp32 operand is carried as two
// 16-bit pieces, hi = round16(x), lo = round16(x - hi) (fp16 in the forward chain, bf16 in the backward chain), and every product as
//      w*x  ~=  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi            (fp32 accumulate, dropped term <= 2^-18 |w x|)
// on v_mfma_f32_16x16x32_{f16,bf16} (both kernels since round 4; rounds 1-2 ran on 32x32x16): 3 MFMAs at the 16-bit rate replace the
// fp32 MFMAs at 1/16 of it, at ~2^-17 (bf16) / 2^-22 (fp16) relative operand error, inside the path's tolerance (PSNR 0.01 dB, depth 1e-4 m).
//
// Structure:
//   * weights are stored as the lane-linear LDS image [k32-step][16-row tile][hi/lo][lane][8 x 16 bit] -> A fragments are plain
//     conflict-free ds_read_b128 at immediate offsets; staging is LDS-DMA into a 3-deep ring with counted vmcnt
//     (two 32 KiB chunks in flight across raw s_barriers);
//   * TWO accumulator sets: layer l+1 consumes operand steps made from layer l's finished accumulators, so layer l's epilogue (ReLU / mask /
//     latent add / hi-lo split) is spread over layer l+1's steps; the last step of a layer deposits the finished sums in the dead
//     previous set;
//   * biases, latent terms and the two small heads are staged in LDS once, so no compiler-tracked global load
//     drains the DMA queue inside the layer chain.
// Limits of this variant (the fp32 kernels have none of them): shape_blocks + texture_blocks <= 4 and
// whole 32-point wave tiles per object.
#include <utility>
#include "snr_mlp_core.hpp"
#include "snr_host.hpp"

namespace snr {
namespace bf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct XOp { bf16x8 hi, lo; };       // one k16-step of a B operand: 8 VGPRs

constexpr int NBUF = 3;
constexpr int WB_BYTES = BF_CHUNK_VIEW;                    // 36864
constexpr int MAX_LAYERS = 8;                              // sb + tb + 4
constexpr int MAX_LAT = 4;
// LDS map (bytes)
constexpr int OFF_VEC = NBUF * WB_BYTES;                   // bias[8][256] | sigma_w[256] | rgb2_w[384] | sigma_b, rgb2_b[3], pad
constexpr int VEC_BIAS = 0, VEC_SIGW = MAX_LAYERS * 256, VEC_RGBW = VEC_SIGW + 256, VEC_MISC = VEC_RGBW + 384;
constexpr int VEC_ZERO = VEC_MISC + 8;                     // 256 zeros: "no latent term" / "no density head" without a branch
constexpr int VEC_FLOATS = VEC_ZERO + 256;
constexpr int OFF_LAT = OFF_VEC + VEC_FLOATS * 4;          // [4 waves][MAX_LAT][256] f32
constexpr int OFF_COMP = OFF_LAT + 4 * MAX_LAT * 256 * 4;  // 128 points x 8 floats
constexpr int OFF_XDIR = OFF_COMP + 128 * COMP_STRIDE * 4; // direction operand steps: [wave][step 2][plane 2][lane 64] x 16 B
constexpr int LDS_BYTES = OFF_XDIR + 4 * 4 * 1024;
constexpr int PE_ROWF = 65;                                // floats per point row of the PE scratch (aliases ring buffer 2)
constexpr int OFF_PE = 2 * WB_BYTES;
static_assert(4 * 32 * PE_ROWF * 4 <= WB_BYTES, "PE scratch must fit in one ring buffer");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(OFF_LAT % 16 == 0 && OFF_COMP % 16 == 0, "alignment");

// ------------------------------------------------------------------------------------------ weight ring
// Three 32 KiB chunks in LDS: chunk ci is being read, ci+1 is landing or has landed, ci+2 ("pending") is being requested, eight
// 1 KiB DMA instructions per wave, a few at a time under the MFMAs that consume chunk ci.
struct Ring {
    const char* next;     // global address of the chunk after the pending one
    int ci;               // index of the next chunk to consume
    int total;            // chunks in the stream
    int use, fill;        // ring buffer holding chunk ci / receiving the next pending chunk
    unsigned wave_lds;    // wave id * 8192 (SGPR): this wave's 8 KiB slice of every chunk
    const char* pg;       // pending chunk: its global address + this wave's slice
    unsigned pm0;         // M0 for its pieces: LDS address of the slice + 4096 (the pieces use immediate offsets -4096 .. +3072)
    const char* last;     // address of the stream's last chunk (fetches past the end re-read it into a vacated buffer)
};
// LDS-DMA through inline asm on purpose: hipcc then does not track these loads, so it cannot put a vmcnt(0) in front of the next
// ds_read "that might alias" (which would drain the prefetch every step); completion is counted by hand in ring_acquire
// (vmcnt(8) + s_barrier before the reads).  A burst of 8 DMA instructions per wave keeps the wave (and, the four waves sharing the CU's address path,
// the whole workgroup) out of the matrix pipe for ~500 cycles per chunk; one bare instruction between MFMAs costs ~20
// (tools/_diag/mfma_order.hip).  Bare = no M0 save/restore and no address arithmetic between pieces: wave w copies the contiguous
// 8 KiB slice w of the chunk, M0 points into the middle of its LDS image and the instruction's immediate offset, which moves the
// global and the LDS address together, selects the KiB.  Nothing else in these kernels uses M0 (checked in the disassembly).
template <int K>
__device__ __forceinline__ void ring_piece(const Ring& r, unsigned voff) {
#ifndef SNR_EXP_NODMA
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :: "v"(voff), "s"(r.pg), "s"(r.pm0), "n"(1024 * K - 4096) : "memory");
#endif
}
template <int K0, int N>
__device__ __forceinline__ void ring_pieces(const Ring& r, unsigned voff) {
    if constexpr (N > 0) { ring_piece<K0>(r, voff); ring_pieces<K0 + 1, N - 1>(r, voff); }
}
// pieces of the pending chunk that belong to position PP of the 2*SPC half-steps between two acquires (8 pieces over the period)
template <int PP, int SPC>
__device__ __forceinline__ void ring_pieces_at(const Ring& r, unsigned voff) {
    constexpr int K0 = PP * 8 / (2 * SPC), N = (PP + 1) * 8 / (2 * SPC) - K0;
    ring_pieces<K0, N>(r, voff);
}
// the same inside the stream's LAST layer of NCH chunks: period W (0 = after the layer's first acquire) requests chunk W + 2 of the layer
template <int PP, int SPC, int W, int NCH, bool TAIL>
__device__ __forceinline__ void ring_pieces_in(const Ring& r, unsigned voff) {
    if constexpr (!TAIL || W + 2 < NCH) ring_pieces_at<PP, SPC>(r, voff);
}
__device__ __forceinline__ void ring_start(Ring& r, const char* stream, int total, char* lds, unsigned voff) {
    r.ci = 0; r.total = total; r.use = 0; r.fill = 2 % NBUF;
    r.wave_lds = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 8192u);
    r.last = stream + (size_t)(total - 1) * BF_CHUNK;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds) + r.wave_lds + 4096u;
    r.pg = stream + r.wave_lds;                r.pm0 = base;             ring_pieces<0, 8>(r, voff);
    r.pg = stream + BF_CHUNK + r.wave_lds;     r.pm0 = base + WB_BYTES;  ring_pieces<0, 8>(r, voff);
    r.next = stream + 2 * BF_CHUNK;
}
// Wait for chunk ci (the 8 youngest DMA instructions are chunk ci+1's), rendezvous, and make chunk ci+2 the pending one; the caller
// issues its 8 pieces before the next acquire.  No branches; the stream's last layer (rgb.0 forward, enc_xyz^T backward: the narrow
// instances of the layer templates) knows at compile time which of its periods have no chunk left to request and that its final
// acquire has nothing younger behind it, so no DMA is in flight when the layers are done.  (`last` only guards a miscounted stream.)
template <bool LAST = false>      // LAST: the stream's final chunk, no younger DMA behind it
__device__ __forceinline__ const char* ring_acquire(Ring& r, char* lds) {
#ifndef SNR_EXP_NOSYNC
    if constexpr (LAST) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
#ifdef SNR_EXP_SAMECHUNK      /* timing experiment: every fetch re-reads the stream's last chunk (L2-hot) */
    const char* nx = r.last;
#else
    const char* nx = r.next < r.last ? r.next : r.last;
#endif
    r.pg = nx + r.wave_lds;
    r.pm0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds) + r.fill * WB_BYTES + r.wave_lds + 4096u;
    r.next = nx + BF_CHUNK;
    r.fill = (r.fill == NBUF - 1) ? 0 : r.fill + 1;
    const char* p = lds + r.use * WB_BYTES;
    r.use = (r.use == NBUF - 1) ? 0 : r.use + 1;
    r.ci += 1;
    return p;
}

// ------------------------------------------------------------------------------------------ operand pieces
__device__ __forceinline__ void split_store(float v, XOp& o, int j) {
    const __bf16 hi = (__bf16)v;
    o.hi[j] = hi;
    o.lo[j] = (__bf16)(v - (float)hi);
}

// pin an operand step where it is produced: without this LLVM sinks the whole epilogue down to its use in the next
// step (behind the barrier), which serialises it with that step's MFMAs instead of hiding it under this step's
__device__ __forceinline__ void pin(XOp& o) { asm volatile("" : "+v"(o.hi), "+v"(o.lo)); }

// ------------------------------------------------------------------------------------------ forward kernel (v_mfma_f32_16x16x32_bf16)
// Round 3: the forward chain runs on the 16x16x32 shape.  Same structure as before -- transposed GEMMs Y^T = W X^T, the accumulators of
// layer l are the B operands of layer l+1 without data movement, weights through the LDS-DMA ring, the previous layer's epilogue spread
// under the MFMAs -- on tiles of 16 features x 16 points: the wave's 32 points are two column blocks c that share every A fragment;
// register r of lane (n = lane & 15, g = lane >> 4) of accumulator tile T of block c holds feature 16 T + 4 g + r of point 16 c + n, and
// an operand step of 32 k takes the tiles 2S and 2S+1 (element j of the lane's 8: feature 32 S + 16 (j >> 2) + 4 g + (j & 3), the k order
// of the forward weight image).  Why: in MFMA-dense loops on random data the chip holds a ~17 % higher clock on this shape (2.15-2.29 vs
// 1.82-1.96 GHz in-kernel, tools/_diag/shape_bench.hip) at 11 % more cycles for the same layer body (twice the MFMA instructions, each
// leaving 8 instead of 24 cycles of issue shadow): the isolated layer chain ran 7-8 % faster in wall time.
// Element type of the FORWARD chain's split operands (activations, forward weight stream): IEEE half.  Two fp16 pieces keep 22 bits of an
// fp32 operand where two bf16 pieces keep 16 (tools/_diag/split_accuracy.py: operand error of a product 3e-7 against 4.5e-6 relative,
// the level of fp32 accumulation itself), at the same three MFMAs per product and +1.4 % of kernel time (0.4929 against 0.4862 ms):
// rendered colours 6e-8 from the exact-fp32 kernel's instead of 2.2e-7 -- and a 60-step training run ends where the reference's own fp32
// arithmetic ends (loss curve 2.1e-6 / weights 1.21e-3 of their movement from the float64 run; bf16 pieces: 3.9e-5 / 1.1e-2, because a
// pre-activation that lands on the other side of zero flips a ReLU).  Activations are clamped to the fp16 range (+-65504) in the same
// v_med3 that applies the ReLU; what lies below 6e-8 in magnitude is lost (absolute, harmless beside biases of order 0.1).  The backward
// chain and the weight-gradient products carry GRADIENTS, whose magnitudes need bf16's exponent range: they stay on bf16 pieces.
// -DSNR_FWD_BF16 builds the bf16 forward of rounds 1-2 (A/B timing).
#ifndef SNR_FWD_BF16
#define SNR_FWD_F16 1
typedef _Float16 fwd_t;
#define SNR_MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#else
typedef __bf16 fwd_t;
#define SNR_MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#endif
typedef fwd_t fwdx8 __attribute__((ext_vector_type(8)));
struct FOp { fwdx8 hi, lo; };
__device__ __forceinline__ void split_store_f(float v, FOp& o, int j) {
    const fwd_t hi = (fwd_t)v;
    o.hi[j] = hi;
    o.lo[j] = (fwd_t)(v - (float)hi);
}
__device__ __forceinline__ void pin(FOp& o) { asm volatile("" : "+v"(o.hi), "+v"(o.lo)); }
#ifdef SNR_STAMPS
#define SNR_STAMP(i) do { if (lane == 0 && tile32 * 32 < io.n_points) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    reinterpret_cast<unsigned long long*>(io.sigmas)[tile32 * 16 + (i)] = t_; } } while (0)
#else
#define SNR_STAMP(i) do {} while (0)
#endif

// interleave request of the forward's groups: 3 MFMAs, then 3 VALU ops (tools/ab_time.py, two boxes, interleaved rounds: 1:1 0.4940 /
// 0.5289 ms forward / forward with ReLU bits, 3:3 0.4840 / 0.5162, 1:0 0.4937 / 0.5232, 2:2, 3:2, 3:4, 4:3, 4:4, 6:6, 8:8, 12:12 all slower
// than 3:3 -- the solver's freedom inside a request matters as much as the ratio)
#ifndef SNR_IL16_MFMA
#define SNR_IL16_MFMA 3
#endif
#ifndef SNR_IL16_VALU
#define SNR_IL16_VALU 3
#endif
#ifndef SNR_IL16_DS
#define SNR_IL16_DS 0          // (an LDS-read request in the groups: 1 or 2 per 3 or 6 MFMAs measured 1.4 - 5 % slower)
#endif
#define SNR_INTERLEAVE16(N_MFMA)                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < (N_MFMA) / SNR_IL16_MFMA; ++g_) {            \
        __builtin_amdgcn_sched_group_barrier(0x008, SNR_IL16_MFMA, 0);                   \
        if (SNR_IL16_DS) __builtin_amdgcn_sched_group_barrier(0x100, SNR_IL16_DS, 0);    \
        if (SNR_IL16_VALU) __builtin_amdgcn_sched_group_barrier(0x002, SNR_IL16_VALU, 0);\
    }

// the backward's groups (bf16_bwd16_kernel) take the same form of request; measured (tools/ab_time.py, two boxes, interleaved rounds, ms):
// 1:1 0.5879 / 0.6029, 1:0 0.5882, 1:2 0.6036, 2:3 0.5926, 3:3 0.5960 / 0.6109, 1:3 0.5992, 2:2 0.6127, 3:2 0.6154, 4:4 0.6191, 1:4 0.6016,
// 2:1 0.6024, 3:4 0.6209, 6:6 0.6250; the 32x32x16 kernel of rounds 1-3 on the same boxes: 0.6031 / 0.6198
#ifndef SNR_IL16B_MFMA
#define SNR_IL16B_MFMA 1
#endif
#ifndef SNR_IL16B_VALU
#define SNR_IL16B_VALU 1
#endif
#define SNR_INTERLEAVE16B(N_MFMA)                                                        \
    _Pragma("unroll") for (int g_ = 0; g_ < (N_MFMA) / SNR_IL16B_MFMA; ++g_) {           \
        __builtin_amdgcn_sched_group_barrier(0x008, SNR_IL16B_MFMA, 0);                  \
        if (SNR_IL16B_VALU) __builtin_amdgcn_sched_group_barrier(0x002, SNR_IL16B_VALU, 0);\
    }

struct Frag16 { fwdx8 hi[4], lo[4]; };        // A fragments of four 16-row tiles (one group): 8 KiB of the chunk, contiguous
__device__ __forceinline__ void load16(Frag16& f, const char* wq /* chunk + group offset + lane*16 */) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f.hi[t] = *reinterpret_cast<const fwdx8*>(wq + (2 * t) * 1024);
        f.lo[t] = *reinterpret_cast<const fwdx8*>(wq + (2 * t + 1) * 1024);
    }
}
// 24 MFMAs of one group: tiles T0 .. T0+3, both column blocks, three split products each.  TO_P: the layer's last step, the finished
// sums go to the (dead) previous accumulator set.
template <int T0, bool TO_P>
__device__ __forceinline__ void mma16(f32x4 (&accC)[2][16], f32x4 (&accP)[2][16], const FOp (&x)[2], const Frag16& f) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 a = SNR_MFMA16(f.hi[t], x[c].hi, accC[c][T0 + t], 0, 0, 0);
            a = SNR_MFMA16(f.hi[t], x[c].lo, a, 0, 0, 0);
            a = SNR_MFMA16(f.lo[t], x[c].hi, a, 0, 0, 0);
            if (TO_P) accP[c][T0 + t] = a; else accC[c][T0 + t] = a;
        }
}
// one whole k32-step of NT16 tiles straight from a chunk (no fragment pipelining): enc_xyz and enc_viewdir's direction step
template <int NT16>
__device__ __forceinline__ void step_mma16(f32x4 (&acc)[2][16], const FOp (&x)[2], const char* ws /* chunk + step offset + lane*16 */) {
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        const fwdx8 ah = *reinterpret_cast<const fwdx8*>(ws + (2 * t) * 1024);
        const fwdx8 al = *reinterpret_cast<const fwdx8*>(ws + (2 * t + 1) * 1024);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            acc[c][t] = SNR_MFMA16(ah, x[c].hi, acc[c][t], 0, 0, 0);
            acc[c][t] = SNR_MFMA16(ah, x[c].lo, acc[c][t], 0, 0, 0);
            acc[c][t] = SNR_MFMA16(al, x[c].hi, acc[c][t], 0, 0, 0);
        }
    }
}

// the same with the fragments of group q + 1 requested before group q's 24 MFMAs (round 4: enc_xyz took 6.2 k cycles for 3.1 k of MFMAs with
// every fragment requested right in front of its use)
__device__ __forceinline__ void step_mma16_groups(f32x4 (&acc)[2][16], const FOp (&x)[2], const char* ws) {
    Frag16 fa, fb;
    load16(fa, ws);
    load16(fb, ws + 8192);
    __builtin_amdgcn_sched_barrier(0);
    mma16<0, false>(acc, acc, x, fa);
    __builtin_amdgcn_sched_barrier(0);
    load16(fa, ws + 16384);
    __builtin_amdgcn_sched_barrier(0);
    mma16<4, false>(acc, acc, x, fb);
    __builtin_amdgcn_sched_barrier(0);
    load16(fb, ws + 24576);
    __builtin_amdgcn_sched_barrier(0);
    mma16<8, false>(acc, acc, x, fa);
    __builtin_amdgcn_sched_barrier(0);
    mma16<12, false>(acc, acc, x, fb);
}

// What happens to a finished 16x16 accumulator tile of the PREVIOUS layer on its way into an operand step of the current one.
struct Epi16 {
    int floor;            // ReLU as an integer max on the bit pattern: 0 for a ReLU layer, INT_MIN for none
    const float* bias;    // LDS: bias of the layer being accumulated (its accumulators start from it)
    const float* zl;      // LDS: latent term added after the activation (a block of zeros if none)
    float* dump[2];       // DUMP (training): the lane's rows of the activation dump for its two points, [point][256] + 4 g, or null
};
// four values (features 16 T + 4 g .. +3 of point 16 c + n) -> elements 4*HALF .. +3 of the operand step; ReLU bits shifted into `mbits`
// in arrival order (T ascending, e ascending; see store_masks16)
template <int HALF, bool MASKS, bool ZADD, bool DUMP>
__device__ __forceinline__ void epi16(const f32x4& acc, FOp& o, const Epi16& c, int T, int cblk, int g, uint32_t& mbits) {
#ifdef SNR_EXP_NOEPI
    return;
#endif
#ifdef SNR_EXP_EPI0      /* timing experiment: the operand step is a raw copy of accumulator bits (one move per two values) */
    { uint32_t (&hw0)[4] = reinterpret_cast<uint32_t (&)[4]>(o.hi); uint32_t (&lw0)[4] = reinterpret_cast<uint32_t (&)[4]>(o.lo);
      hw0[2 * HALF] = __float_as_uint(acc[0]); hw0[2 * HALF + 1] = __float_as_uint(acc[1]); lw0[2 * HALF] = __float_as_uint(acc[2]); lw0[2 * HALF + 1] = __float_as_uint(acc[3]);
      if (HALF == 1) pin(o); return; }
#endif
    f32x4 z;
    if (ZADD) z = *reinterpret_cast<const f32x4*>(c.zl + 16 * T + 4 * g);
    f32x4 dv;
#ifdef SNR_FWD_F16
    // ReLU and the clamp to the fp16 range in one v_med3 (floor = -65504 for the layer without an activation); the pieces are packed two
    // at a time with v_cvt_pkrtz_f16_f32 (the remainder x - hi is exact in fp32 whichever way hi was rounded)
    const float lo_bound = c.floor == 0 ? 0.f : -65504.f;
    float xv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = acc[e];           // bias already inside
        const float y = __builtin_amdgcn_fmed3f(a, lo_bound, 65504.f);
        // the saved bit is "inactive" = (y == 0) = top bit of (bits(y) - 1) for the ReLU's y >= 0: a pre-activation of exactly +0.0 is
        // inactive like torch.relu's derivative at 0 (the sign bit of `a` alone, round 3, called +0.0 active: dead units with zero bias)
        if (MASKS) mbits = __builtin_amdgcn_alignbit(mbits, __builtin_bit_cast(uint32_t, y) - 1u, 31);
        xv[e] = ZADD ? y + z[e] : y;
        if (DUMP) dv[e] = xv[e];
    }
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    float one_ = 1.0f;
    asm("" : "+v"(one_));          // (opaque: a literal 1.0 would be folded to a subtraction, and the conversion would come back)
    uint32_t (&hw)[4] = reinterpret_cast<uint32_t (&)[4]>(o.hi);
    uint32_t (&lw)[4] = reinterpret_cast<uint32_t (&)[4]>(o.lo);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const h2 hp = __builtin_amdgcn_cvt_pkrtz(xv[2 * k], xv[2 * k + 1]);
        // the remainder x - hi as ONE v_fma_mix_f32 (f32 x 1 - f16 piece; exact) instead of v_cvt_f32_f16 + v_sub_f32: three VALU
        // instructions per value in this epilogue where bf16 pieces took four (forward -1.5 %, forward with ReLU bits -2.7 %)
#ifdef SNR_EXP_EPI1      /* timing experiment: no low piece (ReLU + one packed conversion per two values) */
        const h2 lp = hp;
        hw[2 * HALF + k] = __builtin_bit_cast(uint32_t, hp);
        lw[2 * HALF + k] = __builtin_bit_cast(uint32_t, lp);
#else
        const h2 lp = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf(xv[2 * k], one_, -(float)hp[0]), __builtin_fmaf(xv[2 * k + 1], one_, -(float)hp[1]));
        hw[2 * HALF + k] = __builtin_bit_cast(uint32_t, hp);
        lw[2 * HALF + k] = __builtin_bit_cast(uint32_t, lp);
        // (round 4, measured and dropped: the low piece as v_fma_mixlo_f16 / v_fma_mixhi_f16 through inline asm -- 2.5 instead of 3 VALU per value,
        // rgb 4.7e-8 instead of 5.5e-8 from the fp32 kernel's -- ran +0.6 % / +1.6 % (forward / with ReLU bits): the solver does not place asm
        // statements under the MFMAs, and the high half is a read-modify-write of the low half's register)
#endif
    }
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = acc[e];           // bias already inside
        const float y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), c.floor));
        if (MASKS) mbits = __builtin_amdgcn_alignbit(mbits, __builtin_bit_cast(uint32_t, y) - 1u, 31);      // inactive = (y == 0), see the fp16 branch
        const float xv = ZADD ? y + z[e] : y;
        split_store_f(xv, o, 4 * HALF + e);
        if (DUMP) dv[e] = xv;
    }
#endif
    if (DUMP) { if (c.dump[cblk]) *reinterpret_cast<f32x4*>(c.dump[cblk] + 16 * T) = dv; }
    if (HALF == 1) pin(o);
    if (MASKS) asm volatile("" : "+v"(mbits));
}

// ---- hand-placed group of the 256-wide forward layers (round 4, VERDICT r3 #2's second lever at source level): the 24 MFMAs of a group
// with exactly one other instruction in (nearly) every gap -- the next group's eight fragment reads on the even gaps, the conversion of one
// accumulator tile of the previous layer (12 - 25 single-instruction statements) spread over the others, the two DMA pieces behind MFMAs 16
// and 19 -- each gap closed by sched_barrier(0), so the compiler keeps the order as written (its own scheduling inside a gap is one or two
// instructions).  Budget per gap (MI355X_MICROARCH.md): the MFMA holds the vector issue 8 of its 16 cycles, a VALU / LDS instruction costs
// 4: one extra instruction per gap leaves the pipe fed.  -DSNR_NO_PLACED: the sched_group_barrier request of round 3 instead (A/B).
struct E16State {
    float y[4], l[4];
    f32x4 z;
    uint32_t hpw[2];
    uint32_t tmp;
    float one_, lo_bound;
};
template <int K, int HALF, bool MASKS, bool ZADD>
__device__ __forceinline__ void e16_op(E16State& s, const f32x4& acc, FOp& o, const Epi16& c, int T, int g, uint32_t& mbits) {
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    constexpr int PV = 1 + 2 * (MASKS ? 1 : 0) + (ZADD ? 1 : 0);
    constexpr int K0 = ZADD ? 1 : 0;              // op 0 of a ZADD conversion: the latent values of the tile
    if constexpr (ZADD && K == 0) { s.z = *reinterpret_cast<const f32x4*>(c.zl + 16 * T + 4 * g); return; }
    constexpr int kk = K - K0;
    if constexpr (kk < 4 * PV) {
        constexpr int e = kk / PV, sub = kk % PV;
        if constexpr (sub == 0) { const float a = acc[e]; s.y[e] = __builtin_amdgcn_fmed3f(a, s.lo_bound, 65504.f); }
        else if constexpr (MASKS && sub == 1) s.tmp = __builtin_bit_cast(uint32_t, s.y[e]) - 1u;
        else if constexpr (MASKS && sub == 2) mbits = __builtin_amdgcn_alignbit(mbits, s.tmp, 31);
        else s.y[e] = s.y[e] + s.z[e];
    } else {
        constexpr int k2 = kk - 4 * PV;
        uint32_t (&hw)[4] = reinterpret_cast<uint32_t (&)[4]>(o.hi);
        uint32_t (&lw)[4] = reinterpret_cast<uint32_t (&)[4]>(o.lo);
        if constexpr (k2 < 2) { const h2 hp = __builtin_amdgcn_cvt_pkrtz(s.y[2 * k2], s.y[2 * k2 + 1]); s.hpw[k2] = __builtin_bit_cast(uint32_t, hp); hw[2 * HALF + k2] = s.hpw[k2]; }
        else if constexpr (k2 < 6) { constexpr int e = k2 - 2; const h2 hp = __builtin_bit_cast(h2, s.hpw[e >> 1]); s.l[e] = __builtin_fmaf(s.y[e], s.one_, -(float)hp[e & 1]); }
        else { constexpr int q = k2 - 6; const h2 lp = __builtin_amdgcn_cvt_pkrtz(s.l[2 * q], s.l[2 * q + 1]); lw[2 * HALF + q] = __builtin_bit_cast(uint32_t, lp); }
    }
}
template <int I, int T0, bool TO_P, bool HAS_F, int PIECE0, bool HAS_E, int HALF, bool MASKS, bool ZADD>
__device__ __forceinline__ void g16_step(f32x4 (&accC)[2][16], f32x4 (&accP)[2][16], const FOp (&x)[2], const Frag16& f, Frag16& fn, const char* wq,
                                         const Ring& ring, unsigned voff, E16State& s, const f32x4& eacc, FOp& eo, const Epi16& c, int T, int g, uint32_t& mbits) {
    constexpr int t = I / 6, cb = (I / 3) % 2, pr = I % 3;
    if constexpr (pr == 0) accC[cb][T0 + t] = SNR_MFMA16(f.hi[t], x[cb].hi, accC[cb][T0 + t], 0, 0, 0);
    else if constexpr (pr == 1) accC[cb][T0 + t] = SNR_MFMA16(f.hi[t], x[cb].lo, accC[cb][T0 + t], 0, 0, 0);
    else if constexpr (TO_P) accP[cb][T0 + t] = SNR_MFMA16(f.lo[t], x[cb].hi, accC[cb][T0 + t], 0, 0, 0);
    else accC[cb][T0 + t] = SNR_MFMA16(f.lo[t], x[cb].hi, accC[cb][T0 + t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);        // (the gap's instruction BEHIND its MFMA: inside one region the compiler puts a load first)
#ifndef SNR_PL_FSTEP
#define SNR_PL_FSTEP 2          /* a fragment read every FSTEP-th gap, from gap SNR_PL_F0 */
#endif
#ifndef SNR_PL_F0
#define SNR_PL_F0 0
#endif
#ifndef SNR_PL_DMA0
#define SNR_PL_DMA0 16
#endif
#ifndef SNR_PL_DMA1
#define SNR_PL_DMA1 19
#endif
#ifndef SNR_PL_SHIFT
#define SNR_PL_SHIFT 1
#endif
    if constexpr (HAS_F && I >= SNR_PL_F0 && I < SNR_PL_F0 + 8 * SNR_PL_FSTEP && ((I - SNR_PL_F0) % SNR_PL_FSTEP) == 0) {
        constexpr int fi = (I - SNR_PL_F0) / SNR_PL_FSTEP, ft = fi / 2, pl = fi % 2;
        if constexpr (pl == 0) fn.hi[ft] = *reinterpret_cast<const fwdx8*>(wq + (2 * ft) * 1024);
        else fn.lo[ft] = *reinterpret_cast<const fwdx8*>(wq + (2 * ft + 1) * 1024);
    }
    if constexpr (PIECE0 >= 0 && I == SNR_PL_DMA0) ring_pieces<(PIECE0 >= 0 ? PIECE0 : 0), 1>(ring, voff);
    if constexpr (PIECE0 >= 0 && I == SNR_PL_DMA1) ring_pieces<(PIECE0 >= 0 ? PIECE0 + 1 : 0), 1>(ring, voff);
    if constexpr (HAS_E) {
        constexpr int NE = (ZADD ? 1 : 0) + 4 * (1 + 2 * (MASKS ? 1 : 0) + (ZADD ? 1 : 0)) + 8;
        constexpr int SHIFT = NE <= 12 ? SNR_PL_SHIFT : 0;          // twelve operations: the odd gaps (the fragment reads have the even ones)
        // operations k with gap(k) == I, gap(k) = k * 24 / NE + SHIFT  (NE <= 25: at most two per gap)
        constexpr int KA = ((I - SHIFT) * NE + 23) / 24;              // first k with k * 24 / NE >= I - SHIFT
        constexpr int KB = ((I - SHIFT + 1) * NE + 23) / 24;          // first k of the next gap
        if constexpr (I - SHIFT >= 0) {
            if constexpr (KA < KB && KA < NE) e16_op<KA, HALF, MASKS, ZADD>(s, eacc, eo, c, T, g, mbits);
            if constexpr (KA + 1 < KB && KA + 1 < NE) e16_op<KA + 1, HALF, MASKS, ZADD>(s, eacc, eo, c, T, g, mbits);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int T0, bool TO_P, bool HAS_F, int PIECE0, bool HAS_E, int HALF, bool MASKS, bool ZADD, int... I>
__device__ __forceinline__ void g16_run(f32x4 (&accC)[2][16], f32x4 (&accP)[2][16], const FOp (&x)[2], const Frag16& f, Frag16& fn, const char* wq,
                                        const Ring& ring, unsigned voff, E16State& s, const f32x4& eacc, FOp& eo, const Epi16& c, int T, int g, uint32_t& mbits,
                                        std::integer_sequence<int, I...>) {
    (g16_step<I, T0, TO_P, HAS_F, PIECE0, HAS_E, HALF, MASKS, ZADD>(accC, accP, x, f, fn, wq, ring, voff, s, eacc, eo, c, T, g, mbits), ...);
}

// One layer: NT16 output tiles (16: a 256-wide layer, one k32-step per 32 KiB chunk; 8: rgb.0, two steps per chunk, the stream's last
// layer) from the 16 tiles of accP (8 operand steps), the sums building up in a local set and deposited in accP by the last step.
// A chunk period is always four groups of four tiles (8 KiB of fragments each): the fragments of group q+1 are fetched while group q's
// 24 MFMAs run, the next chunk is acquired in the period's last group, its successor's eight DMA pieces go out two per group, and the
// previous layer's epilogue for operand step S+1 is spread over the groups of step S.
// mw: the lane's ReLU bits of the layer being consumed, natural order: word 2 c + (T >> 3), see store_masks16.
template <int NT16, bool MASKS, bool ZADD, bool DUMP>
__device__ __forceinline__ void layer16(f32x4 (&accP)[2][16], Ring& ring, char* lds, const Epi16& c, uint32_t (&mw)[4], int lane) {
    f32x4 accC[2][16];
    const int g = lane >> 4;
    const unsigned voff = lane * 16u + 4096u;
    constexpr int GPS = NT16 / 4;                     // groups per step
    constexpr int NG = 8 * GPS;                       // groups of the layer
    constexpr bool TAIL = (NT16 != 16);               // rgb.0: the stream ends with this layer
    constexpr int NCH = NG / 4;                       // chunks of this layer
#pragma unroll
    for (int i = 0; i < 4; ++i) mw[i] = 0u;
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(c.bias + 16 * t + 4 * g);
        accC[0][t] = b; accC[1][t] = b;
    }
    FOp xc[2], xn[2];
    epi16<0, MASKS, ZADD, DUMP>(accP[0][0], xc[0], c, 0, 0, g, mw[0]); epi16<1, MASKS, ZADD, DUMP>(accP[0][1], xc[0], c, 1, 0, g, mw[0]);
    epi16<0, MASKS, ZADD, DUMP>(accP[1][0], xc[1], c, 0, 1, g, mw[2]); epi16<1, MASKS, ZADD, DUMP>(accP[1][1], xc[1], c, 1, 1, g, mw[2]);
    Frag16 fa, fb;
    const char* w = ring_acquire(ring, lds) + lane * 16;
    load16(fa, w);
    if constexpr (!TAIL || 2 < NCH) ring_pieces<0, 2>(ring, voff);
#ifdef SNR_EXP_EPIIND     /* timing experiment: the MFMAs do not wait for the conversions (they multiply the layer's first operand step throughout) */
    FOp x_first[2] = {xc[0], xc[1]};
#define SNR_MMA16_CALL(T0_, LASTS_, FCUR) mma16<T0_, LASTS_>(accC, accP, x_first, FCUR); asm volatile("" :: "v"(xc[0].hi), "v"(xc[0].lo), "v"(xc[1].hi), "v"(xc[1].lo));
#else
#define SNR_MMA16_CALL(T0_, LASTS_, FCUR) mma16<T0_, LASTS_>(accC, accP, xc, FCUR);
#endif
    // group G: step S = G / GPS, tiles 4 (G % GPS); position Q = G % 4 in its chunk W = G / 4
#if !defined(SNR_NO_PLACED)
#define SNR_GROUP16_PLACED(G, FCUR, FNXT)                                                                                  \
    {                                                                                                                      \
        constexpr int S_ = (G) / GPS, T0_ = 4 * ((G) % GPS), Q_ = (G) % 4, W_ = (G) / 4;                                   \
        constexpr bool LASTS_ = S_ == 7;                                                                                   \
        constexpr bool HASF_ = (Q_ != 3) || ((G) + 1 < NG);                                                                \
        const char* wq_ = w + (Q_ + 1) * 8192;                                                                             \
        if constexpr (Q_ == 3 && (G) + 1 < NG) { w = ring_acquire<TAIL && W_ + 1 == NCH - 1>(ring, lds) + lane * 16; wq_ = w; } \
        constexpr int P0_ = (Q_ != 3) ? ((!TAIL || W_ + 2 < NCH) ? 2 * Q_ + 2 : -1) : (((G) + 1 < NG && (!TAIL || W_ + 3 < NCH)) ? 0 : -1); \
        constexpr int e_ = (G) % 4, cb_ = e_ >> 1, hf_ = e_ & 1, T_ = LASTS_ ? 0 : 2 * (S_ + 1) + hf_;                      \
        g16_run<T0_, LASTS_, HASF_, P0_, !LASTS_, hf_, MASKS, ZADD>(accC, accP, xc, FCUR, FNXT, wq_, ring, voff, es, accP[cb_][T_], xn[cb_], c, T_, g, \
                                                                   mw[2 * cb_ + (T_ >> 3)], std::make_integer_sequence<int, 24>{}); \
        if constexpr (!LASTS_ && ((G) % GPS) == GPS - 1) { xc[0] = xn[0]; xc[1] = xn[1]; }                                 \
    }
#endif
#define SNR_GROUP16(G, FCUR, FNXT)                                                                                         \
    {                                                                                                                      \
        constexpr int S_ = (G) / GPS, T0_ = 4 * ((G) % GPS), Q_ = (G) % 4, W_ = (G) / 4;                                   \
        constexpr bool LASTS_ = S_ == 7;                                                                                   \
        if constexpr (Q_ != 3) load16(FNXT, w + (Q_ + 1) * 8192);                                                          \
        else if constexpr ((G) + 1 < NG) { w = ring_acquire<TAIL && W_ + 1 == NCH - 1>(ring, lds) + lane * 16; load16(FNXT, w); } \
        SNR_MMA16_CALL(T0_, LASTS_, FCUR)                                                                                  \
        if constexpr (Q_ != 3) { if constexpr (!TAIL || W_ + 2 < NCH) ring_pieces<2 * Q_ + 2, 2>(ring, voff); }            \
        else if constexpr ((G) + 1 < NG) { if constexpr (!TAIL || W_ + 3 < NCH) ring_pieces<0, 2>(ring, voff); }           \
        if constexpr (!LASTS_) {                                                                                           \
            /* operand step S+1: four tile conversions (c0 h0, c0 h1, c1 h0, c1 h1) over the GPS groups of step S */      \
            constexpr int E0_ = ((G) % GPS) * (4 / GPS);                                                                   \
            _Pragma("unroll") for (int e_ = E0_; e_ < E0_ + 4 / GPS; ++e_) {                                               \
                const int cb_ = e_ >> 1, hf_ = e_ & 1, T_ = 2 * (S_ + 1) + hf_;                                            \
                if (hf_ == 0) epi16<0, MASKS, ZADD, DUMP>(accP[cb_][T_], xn[cb_], c, T_, cb_, g, mw[2 * cb_ + (T_ >> 3)]);  \
                else epi16<1, MASKS, ZADD, DUMP>(accP[cb_][T_], xn[cb_], c, T_, cb_, g, mw[2 * cb_ + (T_ >> 3)]);           \
            }                                                                                                              \
        }                                                                                                                  \
        SNR_INTERLEAVE16(24)                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if constexpr (!LASTS_ && ((G) % GPS) == GPS - 1) { xc[0] = xn[0]; xc[1] = xn[1]; }                                 \
    }
#define SNR_GROUP16_PAIR(G) SNR_GROUP16(G, fa, fb) SNR_GROUP16((G) + 1, fb, fa)
#if !defined(SNR_NO_PLACED)
#define SNR_GROUP16_PPAIR(G) SNR_GROUP16_PLACED(G, fa, fb) SNR_GROUP16_PLACED((G) + 1, fb, fa)
    if constexpr (NT16 == 16 && !DUMP) {
        E16State es;
        es.one_ = 1.0f; asm("" : "+v"(es.one_));
        es.lo_bound = c.floor == 0 ? 0.f : -65504.f;
        SNR_GROUP16_PPAIR(0) SNR_GROUP16_PPAIR(2) SNR_GROUP16_PPAIR(4) SNR_GROUP16_PPAIR(6)
        SNR_GROUP16_PPAIR(8) SNR_GROUP16_PPAIR(10) SNR_GROUP16_PPAIR(12) SNR_GROUP16_PPAIR(14)
        SNR_GROUP16_PPAIR(16) SNR_GROUP16_PPAIR(18) SNR_GROUP16_PPAIR(20) SNR_GROUP16_PPAIR(22)
        SNR_GROUP16_PPAIR(24) SNR_GROUP16_PPAIR(26) SNR_GROUP16_PPAIR(28) SNR_GROUP16_PPAIR(30)
    } else
#undef SNR_GROUP16_PPAIR
#endif
    {
    SNR_GROUP16_PAIR(0) SNR_GROUP16_PAIR(2) SNR_GROUP16_PAIR(4) SNR_GROUP16_PAIR(6)
    SNR_GROUP16_PAIR(8) SNR_GROUP16_PAIR(10) SNR_GROUP16_PAIR(12) SNR_GROUP16_PAIR(14)
    if constexpr (NG == 32) {
        SNR_GROUP16_PAIR(16) SNR_GROUP16_PAIR(18) SNR_GROUP16_PAIR(20) SNR_GROUP16_PAIR(22)
        SNR_GROUP16_PAIR(24) SNR_GROUP16_PAIR(26) SNR_GROUP16_PAIR(28) SNR_GROUP16_PAIR(30)
    }
    }
#undef SNR_GROUP16_PAIR
#undef SNR_GROUP16
#undef SNR_MMA16_CALL
}

// x | x[lane ^ 32] (the two lanes hold disjoint bits): gfx950 v_permlane32_swap, no LDS round trip
__device__ __forceinline__ uint32_t or_halves(uint32_t v) {
    uint32_t a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a | b;
}
// a 16-bit value's four nibbles spread to the even nibbles of a word
__device__ __forceinline__ uint32_t spread_nibbles(uint32_t x) {
    x = (x | (x << 8)) & 0x00FF00FFu;
    return (x | (x << 4)) & 0x0F0F0F0Fu;
}
// The lane's natural ReLU-bit words -> the documented layout (snr_layout.h: per 32-point tile and ReLU layer 64 lanes x uint4, lane
// 32 h + p, bit (T32 & 1) * 16 + r of word T32 >> 1 = unit 32 T32 + 8 (r >> 2) + 4 h + (r & 3)) that the backward kernels and
// tests/relu_bits.py read.  Natural: word mw[2 c + (T >> 3)] received the sign bits of tiles T = 8 q .. 8 q + 7 (four values each) by
// alignbit, first arrival in bit 31: bit i of ~bitreverse(word) = "unit 16 T + 4 g + e is positive" with i = 4 (T & 7) + e.  Unit
// 16 T + 4 g + e of point 16 c + n belongs to documented lane 32 (g & 1) + 16 c + n, word T >> 2, nibble 2 (T & 3) + (g >> 1): this lane
// holds the even or the odd nibbles of every word, lane ^ 32 the others.
template <int NWORDS /* natural words per column block: 2 (256 units) or 1 (128) */>
__device__ __forceinline__ uint4 masks16_to_layout(const uint32_t (&mw)[4], int g) {
    uint32_t out[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t nat = q < NWORDS ? ~__builtin_bitreverse32(mw[2 * c + q]) : 0u;
            const unsigned sh = (g >> 1) * 4;
            out[c][2 * q] = or_halves(spread_nibbles(nat & 0xFFFFu) << sh);
            out[c][2 * q + 1] = or_halves(spread_nibbles(nat >> 16) << sh);
        }
    const bool c1 = (g >> 1) != 0;
    return make_uint4(c1 ? out[1][0] : out[0][0], c1 ? out[1][1] : out[0][1], c1 ? out[1][2] : out[0][2], c1 ? out[1][3] : out[0][3]);
}

template <int MODE, bool MASKS, bool EBIAS, bool DUMP = false>      // EBIAS: io.latent_bias holds the latent terms folded into the next layers' biases
__global__ void __launch_bounds__(256, 1)                           // DUMP (training): io.act receives every MFMA layer's fp32 input
bf16_fwd_kernel(DecoderIO io, Layout L, const float* __restrict__ xyz, const float* __restrict__ viewdir, RayGeom g,
                float* __restrict__ out_rgb, float* __restrict__ out_depth, float* __restrict__ out_acc) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6) /* SGPR: so is all that derives from it */, p = lane & 31, h = lane >> 5;
    const long long tile128 = blockIdx.x;
    const long long tile32 = tile128 * 4 + wave;
    const long long gp_raw = tile128 * 128 + wave * 32 + p;
    const bool live = gp_raw < io.n_points;
    const long long gp = live ? gp_raw : io.n_points - 1;
    const int sb = io.sb, tb = io.tb;
    const int n_relu = n_relu_layers(sb, tb);
    const int li_encshape = sb + 1, li_view = sb + 2, li_last = sb + tb + 2;
    float* vec = reinterpret_cast<float*>(lds + OFF_VEC);
    float* latw = reinterpret_cast<float*>(lds + OFF_LAT) + wave * MAX_LAT * 256;
    SNR_STAMP(0);

    // ---- the small vectors and this wave's latent terms are REQUESTED here (plain 16-byte loads, all issued before the first wait) and written
    // to LDS behind the positional encodings (round 4: 3.5 k cycles of exposed global round trip at the top of the kernel otherwise); the
    // first ring_acquire's barrier publishes them.  The weight ring's DMA pieces go out in between: they are younger than these loads, so a
    // counted wait for these never waits for less than it should.
    const f32x4* src4 = reinterpret_cast<const f32x4*>(io.packed + L.bias);
    const int n4 = L.n_mfma_layers * 64;
    f32x4 bv[MAX_LAYERS * 64 / 256];
#pragma unroll
    for (int k = 0; k < MAX_LAYERS * 64 / 256; ++k) bv[k] = src4[min(tid + 256 * k, n4 - 1)];
    const float st_sigw = io.packed[L.sigma_w + tid];
    const float st_rgbw0 = io.packed[L.rgb2_w + tid], st_rgbw1 = io.packed[L.rgb2_w + 256 + (tid & 127)];
    const float st_misc = io.packed[(tid < 4 ? L.sigma_b : L.rgb2_b - 4) + (tid & 7)];
    f32x4 lv[MAX_LAT];
    {
        const long long first = tile32 * 32 < io.n_points ? tile32 * 32 : io.n_points - 1;
        const f32x4* ls4 = reinterpret_cast<const f32x4*>((EBIAS ? io.latent_bias : io.latent) + (first / io.points_per_obj) * L.n_lat * 256);
#pragma unroll
        for (int la = 0; la < MAX_LAT; ++la) lv[la] = ls4[max(min(la, L.n_lat - 1), 0) * 64 + lane];   // (n_lat == 0: the caller passes one dummy row)
    }
    auto staged_to_lds = [&]() {
        f32x4* dst4 = reinterpret_cast<f32x4*>(vec + VEC_BIAS);
#pragma unroll
        for (int k = 0; k < MAX_LAYERS * 64 / 256; ++k) if (tid + 256 * k < n4) dst4[tid + 256 * k] = bv[k];
        vec[VEC_SIGW + tid] = st_sigw;
        vec[VEC_RGBW + tid] = st_rgbw0;
        if (tid < 128) vec[VEC_RGBW + 256 + tid] = st_rgbw1;
        if (tid < 8) vec[VEC_MISC + tid] = st_misc;
        vec[VEC_ZERO + tid] = 0.f;
#pragma unroll
        for (int la = 0; la < MAX_LAT; ++la) if (la < L.n_lat) reinterpret_cast<f32x4*>(latw)[la * 64 + lane] = lv[la];
    };
    float px, py, pz, dx, dy, dz, zc = 0.f;
    if (MODE == 0) {
        px = xyz[gp * 3]; py = xyz[gp * 3 + 1]; pz = xyz[gp * 3 + 2];
        dx = viewdir[gp * 3]; dy = viewdir[gp * 3 + 1]; dz = viewdir[gp * 3 + 2];
    } else {
        const long long ray = gp / g.S;
        const SamplePoint sp = make_sample(g, ray, (int)(gp - ray * g.S));
        px = sp.x; py = sp.y; pz = sp.z; dx = sp.dx; dy = sp.dy; dz = sp.dz; zc = sp.zc;
        if (lane < 32) reinterpret_cast<float*>(lds + OFF_COMP)[(wave * 32 + p) * COMP_STRIDE + 4] = zc;
    }
    SNR_STAMP(1);

    // ---- weight ring: chunks 0,1 in flight while the positional encodings are computed (scratch = ring buffer 2)
    Ring ring;
    const int total_chunks = 2 + 8 * (sb + 1) + 9 + 8 * tb + 4;
    const unsigned voff = lane * 16u + 4096u;
    ring_start(ring, reinterpret_cast<const char*>(io.packed + L.bf_fwd), total_chunks, lds, voff);

    // Positional encodings: lane (p, h) computes half of point p's sine / cosine pairs into the point's scratch row (rows are private
    // to the wave: LDS operations of one wave execute in order); the operand steps then gather, for lane (n, gq), the features
    // 32 s + 16 (j >> 2) + 4 gq + (j & 3) of the points 16 c + n.
    const int n16 = lane & 15, gq = lane >> 4;
    FOp x0[2][2];                                     // enc_xyz: [k32-step][column block]
    char* xdir = lds + OFF_XDIR + wave * 4096 + lane * 16;
    {
        float* scw = reinterpret_cast<float*>(lds + OFF_PE) + (wave * 32) * PE_ROWF;
        float* sc = scw + p * PE_ROWF;
#pragma unroll 1
        for (int i = 0; i < 14; i += 2) {                  // two (frequency, axis) pairs per trip: packed arithmetic (pe_sincos2)
            const int q = 15 * h + i;
            f32x2 sn, cs;
            pe_sincos2(f32x2{ldexpf(pick3(px, py, pz, q % 3), q / 3), ldexpf(pick3(px, py, pz, (q + 1) % 3), (q + 1) / 3)}, &sn, &cs);
            sc[3 + q] = sn[0]; sc[3 + 3 * XYZ_FREQ + q] = cs[0];
            sc[4 + q] = sn[1]; sc[4 + 3 * XYZ_FREQ + q] = cs[1];
        }
        {
            const int q = 15 * h + 14;
            float sn, cs;
            pe_sincos(ldexpf(pick3(px, py, pz, q % 3), q / 3), &sn, &cs);
            sc[3 + q] = sn; sc[3 + 3 * XYZ_FREQ + q] = cs;
        }
        if (h == 0) { sc[0] = px; sc[1] = py; sc[2] = pz; sc[63] = 0.f; }
        {   // all scratch reads first, then the conversions (order pinned: the compiler otherwise waits for every read on its own)
            float pv[2][2][8];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[s][c][j] = scw[(16 * c + n16) * PE_ROWF + 32 * s + 16 * (j >> 2) + 4 * gq + (j & 3)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 8; ++j) split_store_f(pv[s][c][j], x0[s][c], j);
        }
#pragma unroll 1
        for (int i = 0; i < 6; i += 2) {
            const int q = 6 * h + i;
            f32x2 sn, cs;
            pe_sincos2(f32x2{ldexpf(pick3(dx, dy, dz, q % 3), q / 3), ldexpf(pick3(dx, dy, dz, (q + 1) % 3), (q + 1) / 3)}, &sn, &cs);
            sc[3 + q] = sn[0]; sc[3 + 3 * DIR_FREQ + q] = cs[0];
            sc[4 + q] = sn[1]; sc[4 + 3 * DIR_FREQ + q] = cs[1];
        }
        if (h == 0) {
            sc[0] = dx; sc[1] = dy; sc[2] = dz;
#pragma unroll
            for (int f = D_DIR; f < 32; ++f) sc[f] = 0.f;
        }
        float dvv[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) dvv[c][j] = scw[(16 * c + n16) * PE_ROWF + 16 * (j >> 2) + 4 * gq + (j & 3)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {      // the direction step of enc_viewdir, parked in LDS until that layer: [column block][plane][lane] x 16 B
            FOp d;
#pragma unroll
            for (int j = 0; j < 8; ++j) split_store_f(dvv[c][j], d, j);
            *reinterpret_cast<fwdx8*>(xdir + c * 2048) = d.hi;
            *reinterpret_cast<fwdx8*>(xdir + c * 2048 + 1024) = d.lo;
        }
    }

    staged_to_lds();
    f32x4 accA[2][16];
    uint32_t mw[4];
    float sig_dot[2] = {0.f, 0.f};
    SNR_STAMP(2);

    // ---- enc_xyz: two k32-steps (one chunk each) straight from the encoding -> accA
    {
        const char* w = ring_acquire(ring, lds) + lane * 16;      // (its barrier also retires the scratch rows)
        ring_pieces<0, 8>(ring, voff);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(vec + VEC_BIAS + 16 * t + 4 * gq);
            accA[0][t] = b; accA[1][t] = b;
        }
        step_mma16_groups(accA, x0[0], w);
        w = ring_acquire(ring, lds) + lane * 16;
        ring_pieces<0, 8>(ring, voff);
        step_mma16_groups(accA, x0[1], w);
    }

    SNR_STAMP(3);
    // ---- 256-wide layers: layer li consumes the accumulators of layer li-1 (epilogue fused into its steps)
    auto epi_of = [&](int l) {     // epilogue configuration of MFMA layer l's output
        Epi16 c;
        c.floor = (l != li_encshape) ? 0 : (int)0x80000000;
        const int la = latent_after(l, sb, tb);
        c.bias = (EBIAS && la >= 0) ? latw + la * 256 : vec + VEC_BIAS + (l + 1) * 256;
        c.zl = (!EBIAS && la >= 0) ? latw + la * 256 : vec + VEC_ZERO;
        c.dump[0] = c.dump[1] = nullptr;
        if (DUMP) {      // slot l = the output of layer l after activation and latent add = the input of layer l + 1
            int t = threadIdx.x; asm volatile("" : "+v"(t));
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const long long gpd = tile128 * 128 + wave * 32 + 16 * cb + (t & 15);
                if (gpd < io.n_points) c.dump[cb] = io.act + ((long long)l * io.n_points + gpd) * 256 + 4 * ((t & 63) >> 4);
            }
        }
        return c;
    };
    const bool tile_live = tile32 * 32 < io.n_points;     // the last workgroup may own wave tiles past the end: they store nothing
    // lane id recomputed where it is needed again late (density / colour heads, composite, stores): values derived from the early
    // `lane` would have to stay in VGPRs across the whole layer chain, and the allocator spills them
    auto fresh_lane = [&]() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t & 63; };
    auto store_mask = [&](int l) {   // ReLU bits of layer l (complete once the next layer has consumed all its tiles)
        if (MASKS && l != li_encshape) {
            const int ln = fresh_lane(), g4 = ln >> 4;
            const uint4 m = masks16_to_layout<2>(mw, g4);
            if (tile_live) io.masks[(tile32 * n_relu + relu_slot(l, sb)) * 64 + 32 * (g4 & 1) + 16 * (g4 >> 1) + (ln & 15)] = m;
        }
    };
    auto extra_dir_step = [&]() {      // enc_viewdir: k = 256 .. 287 are the direction features (their operand steps wait in LDS)
        const int ln = fresh_lane();
        const char* w = ring_acquire(ring, lds) + ln * 16;
        ring_pieces<0, 8>(ring, ln * 16u + 4096u);
        const char* xd = lds + OFF_XDIR + wave * 4096 + ln * 16;
        FOp d[2];
        d[0].hi = *reinterpret_cast<const fwdx8*>(xd);        d[0].lo = *reinterpret_cast<const fwdx8*>(xd + 1024);
        d[1].hi = *reinterpret_cast<const fwdx8*>(xd + 2048); d[1].lo = *reinterpret_cast<const fwdx8*>(xd + 3072);
        step_mma16<16>(accA, d, w);
    };
#pragma unroll 1
    for (int li = 1; li <= li_last; ++li) {
        layer16<16, MASKS, !EBIAS, DUMP>(accA, ring, lds, epi_of(li - 1), mw, lane);
        if (li == li_view) extra_dir_step();
        store_mask(li - 1);
        if (li == li_encshape) {      // density head: this lane's share of w_sigma . y on the finished enc_shape tiles (bias included)
            const int g4 = fresh_lane() >> 4;
            const float* wsig = vec + VEC_SIGW;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wsig + 16 * t + 4 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sig_dot[0] = fmaf(wv[e], accA[0][t][e], sig_dot[0]); sig_dot[1] = fmaf(wv[e], accA[1][t][e], sig_dot[1]); }
            }
        }
        SNR_STAMP(3 + li);
    }
    // ---- rgb.0: 256 -> 128 (8 tiles) from the last 256-wide layer's accumulators
    layer16<8, MASKS, !EBIAS, DUMP>(accA, ring, lds, epi_of(li_last), mw, lane);
    store_mask(li_last);
    SNR_STAMP(12);

    const int lane_t = fresh_lane(), n_t = lane_t & 15, g_t = lane_t >> 4;
    // density head: the four lanes (n, 0..3) of a point hold its four shares
    float o_sigma[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const float pre = sum_halves(sum_row_pairs(sig_dot[cb])) + vec[VEC_MISC + 0];
        o_sigma[cb] = pre > 20.f ? pre : log1pf(expf(pre));
    }

    // ---- colour head: ReLU(rgb.0) . W2 on the VALU
    float pr[2] = {0.f, 0.f}, pg[2] = {0.f, 0.f}, pb[2] = {0.f, 0.f};
    {
        uint32_t mk[4] = {0u, 0u, 0u, 0u};          // natural words 2 c (tiles 0..7); same arrival order as the layers' bits
        const float* w2 = vec + VEC_RGBW;
        float* hdump[2] = {nullptr, nullptr};       // training: ReLU(rgb.0), the input of rgb.2, slot li_last + 1 (128 columns)
        if (DUMP) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const long long gpd = tile128 * 128 + wave * 32 + 16 * cb + n_t;
                if (gpd < io.n_points) hdump[cb] = io.act + ((long long)(li_last + 1) * io.n_points + gpd) * 256 + 4 * g_t;
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 wr = *reinterpret_cast<const f32x4*>(w2 + 16 * t + 4 * g_t);
            const f32x4 wg = *reinterpret_cast<const f32x4*>(w2 + 128 + 16 * t + 4 * g_t);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w2 + 256 + 16 * t + 4 * g_t);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x4 dv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = accA[cb][t][e];
                    v = fmaxf(v, 0.f);
                    if (MASKS) mk[2 * cb] = __builtin_amdgcn_alignbit(mk[2 * cb], __builtin_bit_cast(uint32_t, v) - 1u, 31);     // inactive = (relu(v) == 0): +0.0 too
                    dv[e] = v;
                    pr[cb] = fmaf(wr[e], v, pr[cb]); pg[cb] = fmaf(wg[e], v, pg[cb]); pb[cb] = fmaf(wb[e], v, pb[cb]);
                }
                if (DUMP) { if (hdump[cb]) *reinterpret_cast<f32x4*>(hdump[cb] + 16 * t) = dv; }
            }
        }
        if (MASKS) {
            const uint4 m = masks16_to_layout<1>(mk, g_t);
            if (tile_live) io.masks[(tile32 * n_relu + (n_relu - 1)) * 64 + 32 * (g_t & 1) + 16 * (g_t >> 1) + n_t] = m;
        }
    }
    float cr[2], cg[2], cb_[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        cr[cb] = sum_halves(sum_row_pairs(pr[cb])) + vec[VEC_MISC + 4];
        cg[cb] = sum_halves(sum_row_pairs(pg[cb])) + vec[VEC_MISC + 5];
        cb_[cb] = sum_halves(sum_row_pairs(pb[cb])) + vec[VEC_MISC + 6];
    }
    // every lane holds both column blocks' results for its n: lanes 0..31 speak for the wave's 32 points (point = lane)
    const bool c1 = (lane_t >> 4) & 1;
    const float my_sigma = c1 ? o_sigma[1] : o_sigma[0], my_r = c1 ? cr[1] : cr[0], my_g = c1 ? cg[1] : cg[0], my_b = c1 ? cb_[1] : cb_[0];
    const int p_t = lane_t & 31;

    SNR_STAMP(13);
#ifndef SNR_STAMPS
    {   // (the point index again, from the fresh lane id)
        const long long gp_e = tile128 * 128 + wave * 32 + p_t;
        if (gp_e < io.n_points && lane_t < 32) {
            if (io.sigmas) io.sigmas[gp_e] = my_sigma;
            if (io.rgbs) { io.rgbs[gp_e * 3] = my_r; io.rgbs[gp_e * 3 + 1] = my_g; io.rgbs[gp_e * 3 + 2] = my_b; }
        }
    }
#endif
    if (MODE == 1) {
        float* comp = reinterpret_cast<float*>(lds + OFF_COMP);
        if (lane_t < 32) {
            float* c = comp + (wave * 32 + p_t) * COMP_STRIDE;
            c[0] = my_sigma; c[1] = my_r; c[2] = my_g; c[3] = my_b;      // c[4] = composite depth, parked there at the start
        }
        __syncthreads();
        const int S = g.S;
        const int rays_here = 128 / S;
        const bool white = g.flags & SNR_WHITE_BKGD;
        for (int r = wave; r < rays_here; r += 4) {
            const long long ray = tile128 * rays_here + r;
            if (ray >= g.n_rays) break;
            const float* c0 = comp + r * S * COMP_STRIDE;
            RayOut o = composite_ray_fwd(S, lane_t, white, [&](int k, float& s_, float& r_, float& g_, float& b_, float& z_, float& zn_) {
                const float* c = c0 + k * COMP_STRIDE;
                s_ = c[0]; r_ = c[1]; g_ = c[2]; b_ = c[3]; z_ = c[4];
                zn_ = (k < S - 1) ? c[COMP_STRIDE + 4] : 0.f;
            });
            if (lane_t == 0) {
                out_rgb[ray * 3] = o.r; out_rgb[ray * 3 + 1] = o.g; out_rgb[ray * 3 + 2] = o.b;
                out_depth[ray] = o.depth; out_acc[ray] = o.acc;
            }
        }
    }
    SNR_STAMP(14);
}

// ------------------------------------------------------------------------------------------ backward
// DPP row operations of the latent-gradient reduce-scatter (reduce_tiles16_dpp).  Through asm volatile: the statements keep their order,
// which keeps every DPP read >= 2 instructions behind the write of its source (the hazard the compiler would otherwise pad).
#define SNR_DPP_SELF(R, CTRL, BANK) asm volatile("v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:" BANK : "+v"(R))
#define SNR_DPP_FROM(R0, R1, CTRL, BANK) asm volatile("v_add_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:" BANK : "+v"(R0) : "v"(R1))
#ifdef SNR_STAMPS
#define SNR_BSTAMP(i) do { if (io.d_t && lane == 0 && tile32 * 32 < io.n_points) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    reinterpret_cast<unsigned long long*>(io.d_t)[tile32 * 16 + (i)] = t_; } } while (0)
#else
#define SNR_BSTAMP(i) do {} while (0)
#endif

// Round 4: the backward chain on the shape the forward took in round 3 (the chip holds a higher clock on it in MFMA-dense loops, see the
// forward's header; rounds 1-3 ran it on v_mfma_f32_32x32x16_bf16: 0.603 against 0.588 ms, then the mask change below).  G_in = W^T G_out per layer, the finished sums of a layer are the next
// layer's B operands, all eight operand steps of a layer stay in registers (enc_viewdir^T's direction tiles multiply them afterwards),
// the previous layer's ReLU mask + hi/lo split spread under the MFMAs, bf16 pieces (gradients need the exponent range) -- on tiles of
// 16 features x 16 points: the wave's 32 points are two column blocks c; register r of lane (n = lane & 15, g = lane >> 4) of tile T of
// block c = feature 16 T + 4 g + r of point 16 c + n; operand step S (32 k) = tiles 2S, 2S+1.  Chunk = one k32-step of the 16 output
// tiles (32 KiB, four groups of four tiles); the transposed stream is packed for this shape (pack_bf16_kernel, transpose != 0).
struct Frag16B { bf16x8 hi[4], lo[4]; };
__device__ __forceinline__ void load16b(Frag16B& f, const char* wq /* chunk + group offset + lane*16 */) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f.hi[t] = *reinterpret_cast<const bf16x8*>(wq + (2 * t) * 1024);
        f.lo[t] = *reinterpret_cast<const bf16x8*>(wq + (2 * t + 1) * 1024);
    }
}
#define SNR_MFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
// 24 MFMAs of one group.  ZERO: the layer's first step, the sums start from the constant 0; TO_P: its last, they go to the dead previous set.
template <int T0, bool ZERO, bool TO_P>
__device__ __forceinline__ void mma16b(f32x4 (&accC)[2][16], f32x4 (&accP)[2][16], const XOp (&x)[2], const Frag16B& f) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 a = SNR_MFMA16B(f.hi[t], x[c].hi, ZERO ? zero4 : accC[c][T0 + t]);
            a = SNR_MFMA16B(f.hi[t], x[c].lo, a);
            a = SNR_MFMA16B(f.lo[t], x[c].hi, a);
            if (TO_P) accP[c][T0 + t] = a; else accC[c][T0 + t] = a;
        }
}
// one whole k32-step of 16 tiles straight from a chunk (rgb.0^T: its operands come from the colour head, nothing to hide under it)
template <bool ZERO>
__device__ __forceinline__ void step_mma16b(f32x4 (&acc)[2][16], const XOp (&x)[2], const char* ws) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ws + (2 * t) * 1024);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(ws + (2 * t + 1) * 1024);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 a = SNR_MFMA16B(ah, x[c].hi, ZERO ? zero4 : acc[c][t]);
            a = SNR_MFMA16B(ah, x[c].lo, a);
            acc[c][t] = SNR_MFMA16B(al, x[c].hi, a);
        }
    }
}

// the same with the fragments of group q + 1 requested before group q's 24 MFMAs (rgb.0^T: 12.6 k cycles for 6.1 k of MFMAs without)
template <bool ZERO>
__device__ __forceinline__ void step_mma16b_groups(f32x4 (&acc)[2][16], const XOp (&x)[2], const char* ws) {
    Frag16B fa, fb;
    load16b(fa, ws);
    load16b(fb, ws + 8192);
    __builtin_amdgcn_sched_barrier(0);
    mma16b<0, ZERO, false>(acc, acc, x, fa);
    __builtin_amdgcn_sched_barrier(0);
    load16b(fa, ws + 16384);
    __builtin_amdgcn_sched_barrier(0);
    mma16b<4, ZERO, false>(acc, acc, x, fb);
    __builtin_amdgcn_sched_barrier(0);
    load16b(fb, ws + 24576);
    __builtin_amdgcn_sched_barrier(0);
    mma16b<8, ZERO, false>(acc, acc, x, fa);
    __builtin_amdgcn_sched_barrier(0);
    mma16b<12, ZERO, false>(acc, acc, x, fb);
}

// Latent-term gradient of one layer in this layout: the two column blocks are added, then the reduce-scatter over the 16 point lanes of a
// DPP row (A: row_mirror, tiles T | T + 8; B: row_half_mirror, T | T + 4; C, D: quad sums): lane i of row g ends with the sums of tiles
// 8 b3 + 4 b2 + {0..3}, features 16 T + 4 g + r; one lane per quad parks them in LDS.  192 VALU instructions (the 32x32 form: 416).
__device__ __forceinline__ void reduce_tiles16_dpp(const f32x4 (&acc)[2][16], float* __restrict__ out /*LDS, 256*/, int lane) {
    const int i = lane & 15, g = lane >> 4;
    float s[16][4], v[8][4];
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[T][r] = acc[0][T][r] + acc[1][T][r]; asm volatile("" : "+v"(s[T][r])); }
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0x3" : "=&v"(v[T][r]) : "v"(s[T][r]));
            asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(v[T][r]) : "v"(s[T + 8][r]));
        }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) { SNR_DPP_SELF(v[T][r], "row_half_mirror", "0x5"); SNR_DPP_FROM(v[T][r], v[T + 4][r], "row_half_mirror", "0xa"); }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) SNR_DPP_SELF(v[T][r], "quad_perm:[1,0,3,2]", "0xf");
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) SNR_DPP_SELF(v[T][r], "quad_perm:[2,3,0,1]", "0xf");
    if ((i & 3) == 0) {
        const int T0 = 8 * ((i >> 3) & 1) + 4 * ((i >> 2) & 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(out + 16 * (T0 + t) + 4 * g) = f32x4{v[t][0], v[t][1], v[t][2], v[t][3]};
    }
}

struct BwdEpi16 {
    uint32_t m[4];        // ReLU bits of the layer for the lane's two points: bit 8 (T & 3) + 4 c + r of word T >> 2 (all ones: no activation)
    const float* wsig;    // LDS: density-head weights (only below enc_shape: null otherwise)
    float dpre[2];        // d loss / d (pre-softplus density) of the lane's two points
    float* dzl;           // LDS: where this wave parks the layer's latent-term gradient (256 floats), or null
    float* dump[2];       // DUMP (training): the lane's rows of the pre-activation gradient dump, [point][256] + 4 g, or null
};
// four values (features 16 T + 4 g .. +3 of point 16 c + n) -> elements 4 HALF .. +3 of an operand step, the saved ReLU bit applied
template <int HALF, bool DUMP>
__device__ __forceinline__ void bwd_conv16(const f32x4& acc, XOp& o, const BwdEpi16& c, int T, int cblk) {
    f32x4 dv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // (through asm: from the builtin LLVM makes v_and + v_cmp + v_cndmask, three VALU instructions through VCC, out of "bit ? v : 0")
        uint32_t keep;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(c.m[T >> 2]), "n"(8 * (T & 3) + 4 * cblk + e));
        const float a = acc[e];           // (a copy first: __builtin_bit_cast on the vector-element lvalue itself reads element 0)
        const float v = __uint_as_float(__float_as_uint(a) & (uint32_t)keep);
        split_store(v, o, 4 * HALF + e);
        if (DUMP) dv[e] = v;
    }
    if (DUMP) { if (c.dump[cblk]) *reinterpret_cast<f32x4*>(c.dump[cblk] + 16 * T) = dv; }
    if (HALF == 1) pin(o);
}

// One transposed layer: NT16 output tiles (16: a 256-row layer, one k32-step per chunk; 4: enc_xyz^T, four steps per chunk, the stream's
// last layer) from the 16 tiles of accP; the operand steps are made from accP one step ahead of their use and ALL kept (x[8][2]); the sums
// build up in a local set and the last step deposits them in accP.  `ninth` (enc_viewdir^T): the two direction tiles follow from one more
// chunk ([step][tile 2][plane][lane]) against the kept operand steps.
template <int NT16, bool DUMP>
__device__ __forceinline__ void layer_bwd16(f32x4 (&accP)[2][16], f32x4 (&accD)[2][2], XOp (&x)[8][2], Ring& ring, char* lds, const BwdEpi16& c,
                                            bool ninth, int lane) {
    f32x4 accC[2][16];
    const int g = lane >> 4;
    const unsigned voff = lane * 16u + 4096u;
    constexpr int GPS = NT16 / 4;                     // groups per step
    constexpr int NG = 8 * GPS;                       // groups of the layer
    constexpr bool TAIL = (NT16 != 16);               // enc_xyz^T: the stream ends with this layer
    constexpr int NCH = NG / 4;                       // chunks of this layer
    if (c.dzl) reduce_tiles16_dpp(accP, c.dzl, lane);
    if (c.wsig) {       // below enc_shape the density head adds d_pre * w_sigma (one pass on the finished tiles)
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(c.wsig + 16 * T + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) { accP[0][T][e] = fmaf(c.dpre[0], wv[e], accP[0][T][e]); accP[1][T][e] = fmaf(c.dpre[1], wv[e], accP[1][T][e]); }
        }
    }
    bwd_conv16<0, DUMP>(accP[0][0], x[0][0], c, 0, 0); bwd_conv16<1, DUMP>(accP[0][1], x[0][0], c, 1, 0);
    bwd_conv16<0, DUMP>(accP[1][0], x[0][1], c, 0, 1); bwd_conv16<1, DUMP>(accP[1][1], x[0][1], c, 1, 1);
    Frag16B fa, fb;
    const char* w = ring_acquire<TAIL && NCH == 1>(ring, lds) + lane * 16;
    load16b(fa, w);
    if constexpr (!TAIL || 2 < NCH) ring_pieces<0, 2>(ring, voff);
    // group G: step S = G / GPS, tiles 4 (G % GPS); position Q = G % 4 in its chunk W = G / 4
#define SNR_GROUP16B(G, FCUR, FNXT)                                                                                        \
    {                                                                                                                      \
        constexpr int S_ = (G) / GPS, T0_ = 4 * ((G) % GPS), Q_ = (G) % 4, W_ = (G) / 4;                                   \
        if constexpr (Q_ != 3) load16b(FNXT, w + (Q_ + 1) * 8192);                                                         \
        else if constexpr ((G) + 1 < NG) { w = ring_acquire<TAIL && W_ + 1 == NCH - 1>(ring, lds) + lane * 16; load16b(FNXT, w); } \
        mma16b<T0_, S_ == 0, S_ == 7>(accC, accP, x[S_], FCUR);                                                            \
        if constexpr (Q_ != 3) { if constexpr (!TAIL || W_ + 2 < NCH) ring_pieces<2 * Q_ + 2, 2>(ring, voff); }            \
        else if constexpr ((G) + 1 < NG) { if constexpr (!TAIL || W_ + 3 < NCH) ring_pieces<0, 2>(ring, voff); }           \
        if constexpr (S_ != 7) {                                                                                           \
            /* operand step S+1: four tile conversions (c0 h0, c0 h1, c1 h0, c1 h1) over the GPS groups of step S */      \
            constexpr int E0_ = ((G) % GPS) * (4 / GPS);                                                                   \
            _Pragma("unroll") for (int e_ = E0_; e_ < E0_ + 4 / GPS; ++e_) {                                               \
                const int cb_ = e_ >> 1, hf_ = e_ & 1, T_ = 2 * (S_ + 1) + hf_;                                            \
                if (hf_ == 0) bwd_conv16<0, DUMP>(accP[cb_][T_], x[S_ + 1][cb_], c, T_, cb_);                              \
                else bwd_conv16<1, DUMP>(accP[cb_][T_], x[S_ + 1][cb_], c, T_, cb_);                                       \
            }                                                                                                              \
        }                                                                                                                  \
        SNR_INTERLEAVE16B(24)                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }
#define SNR_GROUP16B_PAIR(G) SNR_GROUP16B(G, fa, fb) SNR_GROUP16B((G) + 1, fb, fa)
    SNR_GROUP16B_PAIR(0) SNR_GROUP16B_PAIR(2) SNR_GROUP16B_PAIR(4) SNR_GROUP16B_PAIR(6)
    if constexpr (NG == 32) {
        SNR_GROUP16B_PAIR(8) SNR_GROUP16B_PAIR(10) SNR_GROUP16B_PAIR(12) SNR_GROUP16B_PAIR(14)
        SNR_GROUP16B_PAIR(16) SNR_GROUP16B_PAIR(18) SNR_GROUP16B_PAIR(20) SNR_GROUP16B_PAIR(22)
        SNR_GROUP16B_PAIR(24) SNR_GROUP16B_PAIR(26) SNR_GROUP16B_PAIR(28) SNR_GROUP16B_PAIR(30)
    }
#undef SNR_GROUP16B_PAIR
#undef SNR_GROUP16B
    if (NT16 == 16 && ninth) {
        w = ring_acquire(ring, lds) + lane * 16;
        ring_pieces<0, 8>(ring, voff);
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(w + s2 * 4096 + t * 2048);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(w + s2 * 4096 + t * 2048 + 1024);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    f32x4 a = SNR_MFMA16B(ah, x[s2][cb].hi, accD[cb][t]);
                    a = SNR_MFMA16B(ah, x[s2][cb].lo, a);
                    accD[cb][t] = SNR_MFMA16B(al, x[s2][cb].hi, a);
                }
            }
    }
}

template <int MODE, bool DUMP = false>      // DUMP (training): io.gdump receives the gradient wrt every MFMA layer's pre-activation
__global__ void __launch_bounds__(256, 1)
bf16_bwd16_kernel(BwdIO io, Layout L, const float* __restrict__ xyz, const float* __restrict__ viewdir, RayGeom g) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, h = lane >> 5, n16 = lane & 15, gq = lane >> 4;
    const long long tile128 = blockIdx.x;
    const long long tile32 = tile128 * 4 + wave;
    const long long gp_raw = tile128 * 128 + wave * 32 + p;
    const bool live = gp_raw < io.n_points;
    const long long gp = live ? gp_raw : io.n_points - 1;
    const int sb = io.sb, tb = io.tb;
    const int n_relu = n_relu_layers(sb, tb);
    const int li_encshape = sb + 1, li_view = sb + 2, li_last = sb + tb + 2;
    float* vec = reinterpret_cast<float*>(lds + OFF_VEC);
    SNR_BSTAMP(0);

    // ---- everything that needs an ordinary global load happens before the DMA ring starts
    vec[VEC_SIGW + tid] = io.packed[L.sigma_w + tid];
    vec[VEC_ZERO + tid] = 0.f;
    for (int i = tid; i < 384; i += 256) vec[VEC_RGBW + i] = io.packed[L.rgb2_w + i];
    const bool tile_live = tile32 * 32 < io.n_points;         // wave tiles past the end (last workgroup) read and store nothing
    // ReLU bits of every ReLU layer for the lane's two points 16 c + n, features 16 T + 4 g + r: in the documented layout (snr_layout.h) they
    // are the nibbles 2 (T & 3) + (g >> 1) of word T >> 2 of lane slot 32 (g & 1) + 16 c + n; block 0's go to the low nibble of every byte,
    // block 1's to the high one: bit 8 (T & 3) + 4 c + r of word T >> 2
    uint4 mk[MAX_LAYERS - 1];
    {
        const int sh = 4 * (gq >> 1);
#pragma unroll
        for (int s = 0; s < MAX_LAYERS - 1; ++s) {
            uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
            if (s < n_relu && tile_live) {
                const uint4* row = io.masks + (tile32 * n_relu + s) * 64 + 32 * (gq & 1) + n16;
                a = row[0]; b = row[16];
            }
            mk[s].x = ((a.x >> sh) & 0x0F0F0F0Fu) | (((b.x >> sh) & 0x0F0F0F0Fu) << 4);
            mk[s].y = ((a.y >> sh) & 0x0F0F0F0Fu) | (((b.y >> sh) & 0x0F0F0F0Fu) << 4);
            mk[s].z = ((a.z >> sh) & 0x0F0F0F0Fu) | (((b.z >> sh) & 0x0F0F0F0Fu) << 4);
            mk[s].w = ((a.w >> sh) & 0x0F0F0F0Fu) | (((b.w >> sh) & 0x0F0F0F0Fu) << 4);
        }
    }
    float px_, py_, pz_, dx, dy, dz, tval = 0.f, zc = 0.f, uval = 0.f;
    long long ray = 0;
    if (MODE == 0) {
        px_ = xyz[gp * 3]; py_ = xyz[gp * 3 + 1]; pz_ = xyz[gp * 3 + 2];
        dx = viewdir[gp * 3]; dy = viewdir[gp * 3 + 1]; dz = viewdir[gp * 3 + 2];
    } else {
        ray = gp / g.S;
        const SamplePoint sp = make_sample(g, ray, (int)(gp - ray * g.S));
        px_ = sp.x; py_ = sp.y; pz_ = sp.z; dx = sp.dx; dy = sp.dy; dz = sp.dz; zc = sp.zc; tval = sp.t; uval = sp.u;
    }
    const float sig_gp = io.sigmas[gp];
    float gs = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gzc = 0.f;
    float* comp = reinterpret_cast<float*>(lds + OFF_COMP);
    if (MODE == 0) {
        if (live) {
            gs = io.d_sigmas ? io.d_sigmas[gp] : 0.f;
            if (io.d_rgbs) { gr = io.d_rgbs[gp * 3]; gg = io.d_rgbs[gp * 3 + 1]; gb = io.d_rgbs[gp * 3 + 2]; }
        }
    } else {
        if (lane < 32) comp[(wave * 32 + p) * COMP_STRIDE + 5] = zc;
        __syncthreads();
        const int S = g.S;
        const int rays_here = 128 / S;
        const bool white = g.flags & SNR_WHITE_BKGD;
        for (int r = wave; r < rays_here; r += 4) {
            const long long rr = tile128 * rays_here + r;
            if (rr >= g.n_rays) break;
            float* c0 = comp + r * S * COMP_STRIDE;
            const float* srow = io.sigmas + rr * S;
            const float* crow = io.rgbs + rr * S * 3;
            const float ur = io.d_rgb ? io.d_rgb[rr * 3] : 0.f, ug = io.d_rgb ? io.d_rgb[rr * 3 + 1] : 0.f,
                        ub = io.d_rgb ? io.d_rgb[rr * 3 + 2] : 0.f;
            const float ud = io.d_depth ? io.d_depth[rr] : 0.f, ua = io.d_acc ? io.d_acc[rr] : 0.f;
            if (S <= 64) composite_ray_bwd<1>(S, lane, white, ur, ug, ub, ud, ua,
                [&](int k, float& s_, float& r_, float& g_, float& b_, float& z_, float& zn_) {
                    s_ = srow[k]; r_ = crow[3 * k]; g_ = crow[3 * k + 1]; b_ = crow[3 * k + 2];
                    z_ = c0[k * COMP_STRIDE + 5];
                    zn_ = (k < S - 1) ? c0[(k + 1) * COMP_STRIDE + 5] : 0.f;
                },
                [&](int k, float ds, float dcr, float dcg, float dcb, float dzz) {
                    float* c = c0 + k * COMP_STRIDE;
                    c[0] = ds; c[1] = dcr; c[2] = dcg; c[3] = dcb; c[4] = dzz;
                });
            else composite_ray_bwd<2>(S, lane, white, ur, ug, ub, ud, ua,
                [&](int k, float& s_, float& r_, float& g_, float& b_, float& z_, float& zn_) {
                    s_ = srow[k]; r_ = crow[3 * k]; g_ = crow[3 * k + 1]; b_ = crow[3 * k + 2];
                    z_ = c0[k * COMP_STRIDE + 5];
                    zn_ = (k < S - 1) ? c0[(k + 1) * COMP_STRIDE + 5] : 0.f;
                },
                [&](int k, float ds, float dcr, float dcg, float dcb, float dzz) {
                    float* c = c0 + k * COMP_STRIDE;
                    c[0] = ds; c[1] = dcr; c[2] = dcg; c[3] = dcb; c[4] = dzz;
                });
        }
        __syncthreads();
        if (live) {
            const float* c = comp + (wave * 32 + p) * COMP_STRIDE;
            gs = c[0]; gr = c[1]; gg = c[2]; gb = c[3]; gzc = c[4];
        }
    }
    // the point's four upstream scalars travel to the lanes (n, g) of its column block through its row of the composite scratch (rows are
    // private to the wave from here on: LDS operations of one wave execute in order)
    if (lane < 32) {
        float* c = comp + (wave * 32 + p) * COMP_STRIDE;
        c[0] = gs * (1.f - expf(-sig_gp)); c[1] = gr; c[2] = gg; c[3] = gb;
    }
    float dpre2[2], gr2[2], gg2[2], gb2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const float* c = comp + (wave * 32 + 16 * cb + n16) * COMP_STRIDE;
        dpre2[cb] = c[0]; gr2[cb] = c[1]; gg2[cb] = c[2]; gb2[cb] = c[3];
    }
    __syncthreads();
    SNR_BSTAMP(1);

    Ring ring;
    const int total_chunks = 4 + 8 * tb + 9 + 8 * (sb + 1) + 2;       // enc_viewdir^T: 8 chunks + 1 for its direction tiles
    const unsigned voff = lane * 16u + 4096u;
    ring_start(ring, reinterpret_cast<const char*>(io.packed + L.bf_bwd), total_chunks, lds, voff);

    XOp x[8][2];
    auto mask_words = [&](int slot, uint32_t (&m)[4]) {        // runtime slot out of the register array (static unroll)
        m[0] = m[1] = m[2] = m[3] = 0xffffffffu;
#pragma unroll
        for (int s = 0; s < MAX_LAYERS - 1; ++s) if (s == slot) { m[0] = mk[s].x; m[1] = mk[s].y; m[2] = mk[s].z; m[3] = mk[s].w; }
    };
    // ---- colour head backward on the VALU: g_h = W2^T d_rgb masked by rgb.0's ReLU -> 4 operand steps (128 features = 8 tiles)
    {
        uint32_t m[4];
        mask_words(n_relu - 1, m);
        const float* w2 = vec + VEC_RGBW;
        float* hdump[2] = {nullptr, nullptr};       // rgb.0's G, 128 columns
        if (DUMP) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const long long gpd = tile128 * 128 + wave * 32 + 16 * cb + n16;
                if (gpd < io.n_points) hdump[cb] = io.gdump + ((long long)(li_last + 1) * io.n_points + gpd) * 256 + 4 * gq;
            }
        }
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const f32x4 wr = *reinterpret_cast<const f32x4*>(w2 + 16 * T + 4 * gq);
            const f32x4 wg = *reinterpret_cast<const f32x4*>(w2 + 128 + 16 * T + 4 * gq);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w2 + 256 + 16 * T + 4 * gq);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x4 dv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = wr[e] * gr2[cb] + wg[e] * gg2[cb] + wb[e] * gb2[cb];
                    v = ((m[T >> 2] >> (8 * (T & 3) + 4 * cb + e)) & 1u) ? v : 0.f;
                    dv[e] = v;
                    split_store(v, x[T >> 1][cb], 4 * (T & 1) + e);
                }
                if (DUMP) { if (hdump[cb]) *reinterpret_cast<f32x4*>(hdump[cb] + 16 * T) = dv; }
            }
        }
    }

    f32x4 accA[2][16], accD[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t) accD[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    SNR_BSTAMP(2);
    // ---- rgb.0^T : K = 128 (4 k32-steps, one chunk each) -> accA
    {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char* w = ring_acquire(ring, lds) + lane * 16;
            ring_pieces<0, 8>(ring, voff);
            if (s == 0) step_mma16b_groups<true>(accA, x[s], w); else step_mma16b_groups<false>(accA, x[s], w);
        }
    }

    auto epi_of = [&](int l) {      // how the gradient wrt (output of layer l [+ latent]) becomes the operand of W_l^T
        BwdEpi16 c;
        mask_words(l == li_encshape ? -1 : relu_slot(l, sb), c.m);
        c.wsig = (l == li_encshape) ? vec + VEC_SIGW : nullptr;
        c.dpre[0] = (l == li_encshape) ? dpre2[0] : 0.f;
        c.dpre[1] = (l == li_encshape) ? dpre2[1] : 0.f;
        const int la = latent_after(l, sb, tb);
        c.dzl = (la >= 0 && io.partial) ? reinterpret_cast<float*>(lds + OFF_LAT) + (wave * MAX_LAT + la) * 256 : nullptr;
        c.dump[0] = c.dump[1] = nullptr;
        if (DUMP) {      // slot l = gradient wrt the pre-activation of MFMA layer l
            int t = threadIdx.x; asm volatile("" : "+v"(t));
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const long long gpd = tile128 * 128 + wave * 32 + 16 * cb + (t & 15);
                if (gpd < io.n_points) c.dump[cb] = io.gdump + ((long long)l * io.n_points + gpd) * 256 + 4 * ((t & 63) >> 4);
            }
        }
        return c;
    };
    SNR_BSTAMP(3);
#pragma unroll 1
    for (int li = li_last; li >= 1; --li) {
        layer_bwd16<16, DUMP>(accA, accD, x, ring, lds, epi_of(li), li == li_view, lane);
        SNR_BSTAMP(4 + li_last - li);
    }
    SNR_BSTAMP(11);
    // ---- enc_xyz^T : 256 -> 64 positional-encoding features (four tiles)
    layer_bwd16<4, DUMP>(accA, accD, x, ring, lds, epi_of(0), false, lane);
    SNR_BSTAMP(12);

    // ---- the parked latent-term gradients of this wave tile -> global partials (the ring is idle now)
    if (io.partial && tile_live) {
        const float* dzl = reinterpret_cast<const float*>(lds + OFF_LAT) + wave * MAX_LAT * 256;
        for (int la = 0; la < L.n_lat; ++la)
            *reinterpret_cast<f32x4*>(io.partial + (tile32 * L.n_lat + la) * 256 + lane * 4) = *reinterpret_cast<const f32x4*>(dzl + la * 256 + lane * 4);
    }
    // ---- positional-encoding backward through the scratch rows (they alias ring buffer 2; the ring retired its last DMA at the final
    // acquire): lane (n, g) writes its share of the rows of points 16 c + n, lane (p, h) reads point p's row
    __syncthreads();
    float* scw = reinterpret_cast<float*>(lds + OFF_PE) + (wave * 32) * PE_ROWF;
    float* sc = scw + p * PE_ROWF;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) scw[(16 * cb + n16) * PE_ROWF + 16 * T + 4 * gq + r] = accA[cb][T][r];
    float gx = 0.f, gy = 0.f, gz = 0.f, hx = 0.f, hy = 0.f, hz = 0.f;
    auto pe_grad = [&](int q, int n_freq, float vx, float vy, float vz, float sn, float cs, float& ax, float& ay, float& az) {
        const int a = q % 3, f = q / 3;
        const float v = ldexpf(sc[3 + q] * cs - sc[3 + 3 * n_freq + q] * sn, f);
        ax += a == 0 ? v : 0.f; ay += a == 1 ? v : 0.f; az += a == 2 ? v : 0.f;
    };
#pragma unroll 1
    for (int i = 0; i < 14; i += 2) {                  // two (frequency, axis) pairs per trip: packed arithmetic (pe_sincos2)
        const int q = 15 * h + i;
        f32x2 sn, cs;
        pe_sincos2(f32x2{ldexpf(pick3(px_, py_, pz_, q % 3), q / 3), ldexpf(pick3(px_, py_, pz_, (q + 1) % 3), (q + 1) / 3)}, &sn, &cs);
        pe_grad(q, XYZ_FREQ, px_, py_, pz_, sn[0], cs[0], gx, gy, gz);
        pe_grad(q + 1, XYZ_FREQ, px_, py_, pz_, sn[1], cs[1], gx, gy, gz);
    }
    {
        const int q = 15 * h + 14;
        float sn, cs;
        pe_sincos(ldexpf(pick3(px_, py_, pz_, q % 3), q / 3), &sn, &cs);
        pe_grad(q, XYZ_FREQ, px_, py_, pz_, sn, cs, gx, gy, gz);
    }
    if (h == 0) { gx += sc[0]; gy += sc[1]; gz += sc[2]; }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) scw[(16 * cb + n16) * PE_ROWF + 16 * t + 4 * gq + r] = accD[cb][t][r];
#pragma unroll 1
    for (int i = 0; i < 6; i += 2) {
        const int q = 6 * h + i;
        f32x2 sn, cs;
        pe_sincos2(f32x2{ldexpf(pick3(dx, dy, dz, q % 3), q / 3), ldexpf(pick3(dx, dy, dz, (q + 1) % 3), (q + 1) / 3)}, &sn, &cs);
        pe_grad(q, DIR_FREQ, dx, dy, dz, sn[0], cs[0], hx, hy, hz);
        pe_grad(q + 1, DIR_FREQ, dx, dy, dz, sn[1], cs[1], hx, hy, hz);
    }
    if (h == 0) { hx += sc[0]; hy += sc[1]; hz += sc[2]; }
    gx = sum_halves(gx); gy = sum_halves(gy); gz = sum_halves(gz);
    hx = sum_halves(hx); hy = sum_halves(hy); hz = sum_halves(hz);
    SNR_BSTAMP(13);

    if (MODE == 0) {
        if (live && h == 0) {
            if (io.d_xyz) { io.d_xyz[gp * 3] = gx; io.d_xyz[gp * 3 + 1] = gy; io.d_xyz[gp * 3 + 2] = gz; }
            if (io.d_dir) { io.d_dir[gp * 3] = hx; io.d_dir[gp * 3 + 1] = hy; io.d_dir[gp * 3 + 2] = hz; }
        }
        return;
    }
    ray_grad_tail(g, io.d_rays_o, io.d_rays_d, io.d_t, comp, tile128, ray, gp, live, tval, uval, zc, gx, gy, gz, hx, hy, hz, gzc);
    SNR_BSTAMP(14);
}

// ------------------------------------------------------------------------------------------ packing
// One thread per (k-step, tile, lane, j): writes the hi and the lo element of the layer image
//   [s][tile][plane hi/lo][lane][8].
//   backward (transpose != 0, v_mfma_f32_16x16x32_bf16): k32-steps, 16-row tiles; value = W[k][row], row = 16 tile +
//     (lane & 15) = input feature, k = output feature 32 s + 16 (j >> 2) + 4 (lane >> 4) + (j & 3);
//   forward (transpose == 0, v_mfma_f32_16x16x32_bf16): k32-steps, 16-row tiles; value = W[row][k], row = 16 tile + (lane & 15) = output
//     feature, k = input feature 32 s + 16 (j >> 2) + 4 (lane >> 4) + (j & 3) -- the order in which the 16x16 accumulator tiles 2s, 2s+1
//     re-enter; k_off shifts k (enc_viewdir's direction step: features 256 .. 282 as one step of their own).
__global__ void pack_bf16_kernel(const float* __restrict__ Wt, int n_out, int k_in, int transpose, int n_tiles, int KS, int tile0, int k_off,
                                 __bf16* __restrict__ dst) {
    const long long total = (long long)n_tiles * KS * 64 * 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const int tile = (int)((i >> 9) % n_tiles);
        const int s = (int)((i >> 9) / n_tiles);
        float v = 0.f;
        if (!transpose) {
            const int row = 16 * (tile0 + tile) + (lane & 15);
            const int k = k_off + 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
            if (row < n_out && k < k_in) v = Wt[(long long)row * k_in + k];
        } else {                          // backward: 16-row tiles of input features, k32-steps over the layer's outputs
            const int row = 16 * (tile0 + tile) + (lane & 15);
            const int k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
            if (row < k_in && k < n_out) v = Wt[(long long)k * k_in + row];
        }
        const long long base = (((long long)s * n_tiles + tile) * 2) * 512;      // 16-bit elements; plane stride 512
        if (!transpose) {                                // the forward stream in the forward chain's element type
#ifdef SNR_FWD_F16
            v = fminf(fmaxf(v, -65504.f), 65504.f);      // (a weight beyond the fp16 range saturates instead of becoming an infinity)
#endif
            const fwd_t hi = (fwd_t)v;
            const fwd_t lo = (fwd_t)(v - (float)hi);
            reinterpret_cast<fwd_t*>(dst)[base + lane * 8 + j] = hi;
            reinterpret_cast<fwd_t*>(dst)[base + 512 + lane * 8 + j] = lo;
        } else {
            const __bf16 hi = (__bf16)v;
            const __bf16 lo = (__bf16)(v - (float)hi);
            dst[base + lane * 8 + j] = hi;
            dst[base + 512 + lane * 8 + j] = lo;
        }
    }
}

}  // namespace bf
}  // namespace snr

using namespace snr;

// ---- host side ---------------------------------------------------------------------------------------------
int snr_bf16_supported_(int sb, int tb, long long points_per_obj) {
    return (sb + tb + 4 <= bf::MAX_LAYERS) && (sb + tb <= bf::MAX_LAT) && (points_per_obj % 32 == 0);
}

int snr_bf16_pack_(const float* const* W /* per-point weight tensors in MFMA-layer order */, int sb, int tb, float* packed, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const Layout L = make_layout(sb, tb);
    auto launch = [&](const float* w, int n_out, int k_in, int transpose, int n_tiles, int KS, __bf16* dst, int tile0 = 0, int k_off = 0) {
        const long long total = (long long)n_tiles * KS * 512;
        int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
        bf::pack_bf16_kernel<<<grid, 256, 0, st>>>(w, n_out, k_in, transpose, n_tiles, KS, tile0, k_off, dst);
    };
    const int n_layers = sb + tb + 4;
    // forward stream (16x16x32 image): k32-steps of 16-row tiles; enc_viewdir = 8 steps over the 256 hidden units + one step over the
    // direction features
    char* f = reinterpret_cast<char*>(packed + L.bf_fwd);
    for (int li = 0; li < n_layers; ++li) {
        const bool is_xyz = li == 0, is_view = li == sb + 2, is_rgb0 = li == n_layers - 1;
        const int n_out = is_rgb0 ? 128 : 256;
        const int k_in = is_xyz ? D_XYZ : (is_view ? 256 + D_DIR : 256);
        const int KS = is_xyz ? 2 : 8;
        const int n_tiles = n_out / 16;
        launch(W[li], n_out, is_view ? 256 + D_DIR : k_in, 0, n_tiles, KS, reinterpret_cast<__bf16*>(f));
        f += (long long)n_tiles * 2 * KS * 1024;
        if (is_view) {                                             // the direction features: one more k32-step, a chunk of its own
            launch(W[li], n_out, k_in, 0, n_tiles, 1, reinterpret_cast<__bf16*>(f), 0, 256);
            f += (long long)n_tiles * 2 * 1024;
        }
    }
    if (f - reinterpret_cast<char*>(packed + L.bf_fwd) != L.bf_fwd_bytes) return SNR_E_SHAPE;
    // backward stream: rgb.0^T, texture^T (reverse), enc_viewdir^T, enc_shape^T, shape^T (reverse), enc_xyz^T
    char* b = reinterpret_cast<char*>(packed + L.bf_bwd);
    for (int li = n_layers - 1; li >= 0; --li) {
        const bool is_xyz = li == 0, is_view = li == sb + 2, is_rgb0 = li == n_layers - 1;
        const int n_out = is_rgb0 ? 128 : 256;
        const int k_in = is_xyz ? D_XYZ : (is_view ? 256 + D_DIR : 256);
        const int KS = n_out / 32;                                 // k32-steps over the layer's outputs
        const int n_tiles = is_xyz ? 4 : 16;                       // tiles of 16 input features
        launch(W[li], n_out, k_in, 2, n_tiles, KS, reinterpret_cast<__bf16*>(b));
        b += (long long)n_tiles * 2 * KS * 1024;
        if (is_view) {                                             // the direction features: tiles 16, 17, a chunk of their own
            launch(W[li], n_out, k_in, 2, 2, KS, reinterpret_cast<__bf16*>(b), 16);
            b += 2ll * 2 * KS * 1024;
        }
    }
    if (b - reinterpret_cast<char*>(packed + L.bf_bwd) != L.bf_bwd_bytes) return SNR_E_SHAPE;
    return snr_check_launch_();
}

int snr_bf16_launch_fwd_(int mode, const DecoderIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, float* rgb,
                         float* depth, float* acc, void* stream_) {
    const unsigned grid = (unsigned)((io.n_points + 127) / 128);
    hipStream_t st = (hipStream_t)stream_;
    const bool ebias = mode == 1 && io.latent_bias && L.n_lat > 0;
    if (io.act) {          // training: points mode, ReLU bits and the per-layer input dump
        if (mode != 0 || !io.masks) return SNR_E_ARG;
        bf::bf16_fwd_kernel<0, true, false, true><<<grid, 256, 0, st>>>(io, L, xyz, viewdir, g, rgb, depth, acc);
    } else if (io.masks) {
        if (mode == 0) bf::bf16_fwd_kernel<0, true, false><<<grid, 256, 0, st>>>(io, L, xyz, viewdir, g, rgb, depth, acc);
        else if (ebias) bf::bf16_fwd_kernel<1, true, true><<<grid, 256, 0, st>>>(io, L, xyz, viewdir, g, rgb, depth, acc);
        else bf::bf16_fwd_kernel<1, true, false><<<grid, 256, 0, st>>>(io, L, xyz, viewdir, g, rgb, depth, acc);
    } else {
        if (mode == 0) bf::bf16_fwd_kernel<0, false, false><<<grid, 256, 0, st>>>(io, L, xyz, viewdir, g, rgb, depth, acc);
        else if (ebias) bf::bf16_fwd_kernel<1, false, true><<<grid, 256, 0, st>>>(io, L, xyz, viewdir, g, rgb, depth, acc);
        else bf::bf16_fwd_kernel<1, false, false><<<grid, 256, 0, st>>>(io, L, xyz, viewdir, g, rgb, depth, acc);
    }
    return snr_check_launch_();
}

int snr_bf16_launch_bwd_(int mode, const BwdIO& io, const Layout& L, const float* xyz, const float* viewdir, const RayGeom& g, void* stream_) {
    const unsigned grid = (unsigned)((io.n_points + 127) / 128);
#define SNR_BWD_KERNEL bf::bf16_bwd16_kernel
    if (io.gdump) {
        if (mode != 0) return SNR_E_ARG;
        SNR_BWD_KERNEL<0, true><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, xyz, viewdir, g);
    } else if (mode == 0) SNR_BWD_KERNEL<0><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, xyz, viewdir, g);
    else SNR_BWD_KERNEL<1><<<grid, 256, 0, (hipStream_t)stream_>>>(io, L, xyz, viewdir, g);
#undef SNR_BWD_KERNEL
    return snr_check_launch_();
}
