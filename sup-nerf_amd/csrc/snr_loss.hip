// Loss / metric tail of the optimise iteration for gfx950: the three masked reductions the reference runs right after the
// render (src/optimizer_nuscenes.py:729-744 == src/optimizer_kitti.py:792-812) and the gradient seeds of the render backward, as
// one launch each instead of ~25 elementwise / reduction launches with host round trips in between.  HBM-bound and tiny
// (36 B per ray): one workgroup per object (1024 threads from 2048 rays on: the kernel is a chain of load latencies, four times
// the threads are a quarter of the trips), rays strided over the threads with coalesced loads, deterministic
// wave -> workgroup reduction (DPP row sums + permlane swaps, then 4 partials through LDS).
#include "snr_device.hpp"
#include "snr_host.hpp"

namespace snr {

constexpr float LOSS_EPS = 1e-9f;     // src/optimizer_nuscenes.py:730,733,742

// sum over the workgroup's threads (256 or 1024), result in every thread; `slot` = 16 floats of LDS per call site; fixed association
__device__ __forceinline__ float block_sum(float v, float* slot) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) slot[wave] = v;
    __syncthreads();
    float s = (slot[0] + slot[1]) + (slot[2] + slot[3]);
    if (blockDim.x > 256) s = (s + ((slot[4] + slot[5]) + (slot[6] + slot[7]))) + (((slot[8] + slot[9]) + (slot[10] + slot[11])) + ((slot[12] + slot[13]) + (slot[14] + slot[15])));
    __syncthreads();
    return s;
}

// out[b] = {loss, loss_rgb, loss_occ, mse_fg};  SEEDS: d_rgb, d_acc = d(sum_b upstream_b * loss_b) / d(rgb, acc)
template <bool SEEDS>
__global__ void __launch_bounds__(1024) loss_tail_kernel(const float* __restrict__ rgb, const float* __restrict__ acc,
                                                        const float* __restrict__ tgt, const float* __restrict__ occ,
                                                        long long rays_per_obj, float coef, const float* __restrict__ upstream,
                                                        float* __restrict__ out, float* __restrict__ d_rgb, float* __restrict__ d_acc) {
    __shared__ float red[16];
    const long long obj = blockIdx.x;
    const long long r0 = obj * rays_per_obj;
    // pass 1: the two denominators depend on the occupancy labels alone
    float sa = 0.f, sf = 0.f;
    for (long long i = threadIdx.x; i < rays_per_obj; i += blockDim.x) {
        const float o = occ[r0 + i];
        sa += fabsf(o);
        sf += fmaxf(o, 0.f);
    }
    const float den = block_sum(sa, red) + LOSS_EPS;
    if (SEEDS) {
        const float g = (upstream ? upstream[obj] : 1.f) / den;
        for (long long i = threadIdx.x; i < rays_per_obj; i += blockDim.x) {
            const long long r = r0 + i;
            const float o = occ[r], a = fabsf(o);
            if (d_rgb) {
#pragma unroll
                for (int c = 0; c < 3; ++c) d_rgb[r * 3 + c] = 2.f * (rgb[r * 3 + c] - tgt[r * 3 + c]) * a * g;
            }
            if (d_acc) d_acc[r] = coef * expf(-o * (0.5f - acc[r])) * o * a * g;
        }
        return;
    }
    const float den_fg = block_sum(sf, red) + LOSS_EPS;
    float s_rgb = 0.f, s_occ = 0.f, s_fg = 0.f;
    for (long long i = threadIdx.x; i < rays_per_obj; i += blockDim.x) {
        const long long r = r0 + i;
        const float o = occ[r], a = fabsf(o);
        const float dr = rgb[r * 3] - tgt[r * 3], dg = rgb[r * 3 + 1] - tgt[r * 3 + 1], db = rgb[r * 3 + 2] - tgt[r * 3 + 2];
        const float sq = dr * dr + dg * dg + db * db;
        s_rgb += sq * a;
        s_fg += sq * fmaxf(o, 0.f);
        s_occ += expf(-o * (0.5f - acc[r])) * a;
    }
    const float l_rgb = block_sum(s_rgb, red) / den;
    const float l_occ = block_sum(s_occ, red) / den;
    const float mse_fg = block_sum(s_fg, red) / den_fg;
    if (threadIdx.x == 0) {
        out[obj * 4 + 0] = l_rgb + coef * l_occ;
        out[obj * 4 + 1] = l_rgb;
        out[obj * 4 + 2] = l_occ;
        out[obj * 4 + 3] = mse_fg;
    }
}

}  // namespace snr

using namespace snr;

extern "C" {

int snr_loss_tail_fwd(const float* rgb, const float* acc, const float* rgb_tgt, const float* occ, int64_t n_rays, int64_t rays_per_obj,
                      float loss_occ_coef, float* out, void* stream) {
    if (n_rays == 0) return SNR_OK;
    if (!rgb || !acc || !rgb_tgt || !occ || !out) return SNR_E_ARG;
    if (n_rays < 0 || rays_per_obj < 1 || (n_rays % rays_per_obj) != 0) return SNR_E_SHAPE;
    loss_tail_kernel<false><<<(unsigned)(n_rays / rays_per_obj), rays_per_obj >= 2048 ? 1024 : 256, 0, (hipStream_t)stream>>>(rgb, acc, rgb_tgt, occ, rays_per_obj, loss_occ_coef,
                                                                                                 nullptr, out, nullptr, nullptr);
    return snr_check_launch_();
}

int snr_loss_tail_bwd(const float* rgb, const float* acc, const float* rgb_tgt, const float* occ, int64_t n_rays, int64_t rays_per_obj,
                      float loss_occ_coef, const float* upstream, float* d_rgb, float* d_acc, void* stream) {
    if (n_rays == 0) return SNR_OK;
    if (!rgb || !acc || !rgb_tgt || !occ || (!d_rgb && !d_acc)) return SNR_E_ARG;
    if (n_rays < 0 || rays_per_obj < 1 || (n_rays % rays_per_obj) != 0) return SNR_E_SHAPE;
    loss_tail_kernel<true><<<(unsigned)(n_rays / rays_per_obj), rays_per_obj >= 2048 ? 1024 : 256, 0, (hipStream_t)stream>>>(rgb, acc, rgb_tgt, occ, rays_per_obj, loss_occ_coef,
                                                                                                upstream, nullptr, d_rgb, d_acc);
    return snr_check_launch_();
}

}  // extern "C"
