// The small per-iteration pieces of the test-time optimisation loop (src/optimizer_nuscenes.py:674-783 == src/optimizer_kitti.py:731-866)
// that surround the render, as single launches for gfx950 -- one object of the batch per workgroup -- so that an iteration is ~30
// launches instead of ~350 and neither the host nor a train of 3-microsecond kernels bounds it:
//   * pose_rays:  pose parameters (axis-angle, translation) -> camera-in-object pose, ray origins / unit directions of the pixel
//                 grid, stratified depths (src/optimizer_nuscenes.py:685-699 + get_rays src/utils.py:107-135 + the sphere bounds and
//                 sample_from_rays' depth vector src/utils.py:159-164,468-469), and its backward to the pose parameters;
//   * metric_row: PSNR / depth change / rotation / translation errors of the iteration (:739-765);
//   * adamw_step: the AdamW update of the four parameter groups (:1762-1769), torch.optim.AdamW's arithmetic.
// All HBM-/latency-bound and tiny; the wave reductions use DPP row sums and permlane swaps (snr_device.hpp).
#include "snr_device.hpp"
#include "snr_host.hpp"

namespace snr {

struct Pose { float Rc[9]; float tc[3]; };

// R = I + a K + b K^2 (Rodrigues), a = sin(t)/t, b = (1 - cos t)/t^2, series below t^2 = 1e-8 (driver.axis_angle_to_matrix)
__device__ __forceinline__ void rodrigues(const float v[3], float R[9], float* a_, float* b_, float* t2_) {
    const float x = v[0], y = v[1], z = v[2];
    const float t2 = x * x + y * y + z * z;
    const float t = sqrtf(fmaxf(t2, 1e-24f));
    const bool small = t2 < 1e-8f;
    const float a = small ? 1.f - t2 / 6.f : sinf(t) / t;
    const float b = small ? 0.5f - t2 / 24.f : (1.f - cosf(t)) / fmaxf(t2, 1e-24f);
    const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    float K2[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.f : 0.f) + a * K[i] + b * K2[i];
    *a_ = a; *b_ = b; *t2_ = t2;
}

// camera-in-object pose from the optimised parameters: object pose (R, t) inverted unless the camera pose itself is optimised
__device__ __forceinline__ Pose make_pose(const float* rot_vec, const float* trans_vec, int opt_cam_pose) {
    float R[9], a, b, t2;
    const float v[3] = {rot_vec[0], rot_vec[1], rot_vec[2]};
    rodrigues(v, R, &a, &b, &t2);
    const float t[3] = {trans_vec[0], trans_vec[1], trans_vec[2]};
    Pose p;
    if (opt_cam_pose) {
#pragma unroll
        for (int i = 0; i < 9; ++i) p.Rc[i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) p.tc[i] = t[i];
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) p.Rc[3 * i + j] = R[3 * j + i];
            p.tc[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
        }
    }
    return p;
}

// the pose as the caller holds it: c2w (3,4) row-major = [R | t] (get_rays' argument, src/utils.py:107)
__device__ __forceinline__ Pose pose_from_c2w(const float* c) {
    Pose p;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) p.Rc[3 * i + j] = c[4 * i + j];
        p.tc[i] = c[4 * i + 3];
    }
    return p;
}

// grid (B objects, chunks of RAYS_PER_BLOCK rays).  cam_dirs (B, n, 3) = [(px - cx)/fx, (py - cy)/fy, 1] of the object's pixels (constant over
// the loop).  DIRECT: `pa` is the camera pose c2w (B,3,4) itself (the public get_rays / render_rays_v2 path); else (pa, pb) = (rot_vec, trans_vec).
constexpr int RAYS_PER_BLOCK = 1024;
template <bool DIRECT>
__global__ void __launch_bounds__(256) pose_rays_fwd_kernel(const float* __restrict__ pa, const float* __restrict__ pb,
                                                            const float* __restrict__ cam_dirs, const float* __restrict__ half_diag,
                                                            const float* __restrict__ jitter, long long n, int S, int opt_cam_pose,
                                                            float* __restrict__ cam2opt, float* __restrict__ rays_o,
                                                            float* __restrict__ viewdir, float* __restrict__ z_vals) {
    const long long b = blockIdx.x;
    const Pose p = DIRECT ? pose_from_c2w(pa + 12 * b) : make_pose(pa + 3 * b, pb + 3 * b, opt_cam_pose);
    if (blockIdx.y == 0) {
        if (threadIdx.x < 12 && cam2opt) {
            const int i = threadIdx.x / 4, j = threadIdx.x % 4;
            cam2opt[b * 12 + threadIdx.x] = j < 3 ? p.Rc[3 * i + j] : p.tc[i];
        }
        if (z_vals && threadIdx.x < S) {     // S <= 256 (checked by the launcher)
            // near / far = |camera centre| -/+ diag/2, detached; two-sided linspace like torch.linspace (utils._linspace)
            const float dist = sqrtf(p.tc[0] * p.tc[0] + p.tc[1] * p.tc[1] + p.tc[2] * p.tc[2]);
            const float hd = half_diag[b];
            const float near = dist - hd, far = dist + hd;
            const float hw = (far - near) / (2.f * S);
            const float start = near + hw, end = far - hw;
            const float step = (end - start) / (float)(S > 1 ? S - 1 : 1);
            const int k = threadIdx.x;
            const float lin = (k < S / 2) ? start + step * (float)k : end - step * (float)(S - 1 - k);
            z_vals[b * S + k] = lin + (jitter ? jitter[b * S + k] : 0.f) * hw;
        }
    }
    const long long i0 = (long long)blockIdx.y * RAYS_PER_BLOCK;
    const long long i1 = i0 + RAYS_PER_BLOCK < n ? i0 + RAYS_PER_BLOCK : n;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float* c = cam_dirs + (b * n + i) * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        float w[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) w[r] = __fadd_rn(__fadd_rn(__fmul_rn(cx, p.Rc[3 * r]), __fmul_rn(cy, p.Rc[3 * r + 1])), __fmul_rn(cz, p.Rc[3 * r + 2]));
        // torch.norm(world, dim=-1) on the CPU accumulates the squares as one fma chain in element order (checked against 1e5 vectors): the same
        // chain here makes the rays bit-equal to the reference's for the same pose
        const float nrm = sqrtf(__fmaf_rn(w[2], w[2], __fmaf_rn(w[1], w[1], __fmul_rn(w[0], w[0]))));
        float* vo = viewdir + (b * n + i) * 3;
        float* oo = rays_o + (b * n + i) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) { vo[r] = __fdiv_rn(w[r], nrm); oo[r] = p.tc[r]; }
    }
}

__device__ __forceinline__ float block_sum256(float v, float* slot) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = (slot[0] + slot[1]) + (slot[2] + slot[3]);
    __syncthreads();
    return s;
}

// d(rays_o), d(viewdir) -> d(rot_vec), d(trans_vec); the depths are detached from the pose like the reference's .tolist() (src/utils.py:468)
// DIRECT: (rot_vec, d_rot_vec) are the pose c2w (B,3,4) and its gradient (B,3,4) = [dL/dR | dL/dt]; trans_vec / d_trans_vec unused.
template <bool DIRECT>
__global__ void __launch_bounds__(1024) pose_rays_bwd_kernel(const float* __restrict__ rot_vec, const float* __restrict__ trans_vec,
                                                            const float* __restrict__ cam_dirs, long long n, int opt_cam_pose,
                                                            const float* __restrict__ d_rays_o, const float* __restrict__ d_viewdir,
                                                            const float* __restrict__ d_cam2opt,
                                                            float* __restrict__ d_rot_vec, float* __restrict__ d_trans_vec) {
    // the 12 sums over the rays (dL/dR, dL/dt) meet in LDS: every thread parks its 12 partials, 12 x 16 threads add a 16th of the block's
    // column each in thread order, 12 threads add the 16 results -- a fixed association, two rendezvous (twelve 6-step double shuffles +
    // 24 rendezvous before).  1024 threads from 2048 rays on: a quarter of the trips.  21 -> 16 us at one object.
    __shared__ double part[12][1024];
    __shared__ double part2[12][16];
    const long long b = blockIdx.x;
    const Pose p = DIRECT ? pose_from_c2w(rot_vec + 12 * b) : make_pose(rot_vec + 3 * b, trans_vec + 3 * b, opt_cam_pose);
    // The direction's gradient is projected off the direction, gu - u (u . gu): d_viewdir is dominated by its component ALONG u (points are
    // o + t d), so the projection cancels most of it and fp32 keeps few digits of the rest.  Per-ray math and the sums over the rays run in
    // double (a few dozen flops per ray: nothing on this chip), so this kernel adds no rounding noise of its own to the pose gradient.
    double gRd[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, gtd[3] = {0., 0., 0.};
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const float* c = cam_dirs + (b * n + i) * 3;
        const double cd[3] = {c[0], c[1], c[2]};
        double w[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) w[r] = cd[0] * (double)p.Rc[3 * r] + cd[1] * (double)p.Rc[3 * r + 1] + cd[2] * (double)p.Rc[3 * r + 2];
        const double inv = 1.0 / sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double u[3] = {w[0] * inv, w[1] * inv, w[2] * inv};
        double gu[3] = {0., 0., 0.};
        if (d_viewdir) { const float* g = d_viewdir + (b * n + i) * 3; gu[0] = g[0]; gu[1] = g[1]; gu[2] = g[2]; }
        const double dot = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double gw = (gu[r] - u[r] * dot) * inv;
#pragma unroll
            for (int j = 0; j < 3; ++j) gRd[3 * r + j] += gw * cd[j];
        }
        if (d_rays_o) { const float* g = d_rays_o + (b * n + i) * 3; gtd[0] += g[0]; gtd[1] += g[1]; gtd[2] += g[2]; }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) part[i][threadIdx.x] = gRd[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) part[9 + i][threadIdx.x] = gtd[i];
    __syncthreads();
    if (threadIdx.x < 12 * 16) {
        const int q = threadIdx.x >> 4, seg = threadIdx.x & 15, len = blockDim.x >> 4;
        double sum = 0.;
        for (int k = 0; k < len; ++k) sum += part[q][seg * len + k];
        part2[q][seg] = sum;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float gR[9], gt[3];
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        double sum = 0.;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += part2[q][k];
        if (q < 9) gR[q] = (float)sum; else gt[q - 9] = (float)sum;
    }
    if (DIRECT) {         // the pose is the leaf: its gradient is [sum_i gw_i cd_i^T | sum_i d_rays_o_i]
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) d_rot_vec[b * 12 + 4 * i + j] = gR[3 * i + j];
            d_rot_vec[b * 12 + 4 * i + 3] = gt[i];
        }
        return;
    }
    if (d_cam2opt) {      // a caller that also used the (3,4) pose itself
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) gR[3 * i + j] += d_cam2opt[b * 12 + 4 * i + j];
            gt[i] += d_cam2opt[b * 12 + 4 * i + 3];
        }
    }
    // gradient wrt the object-pose rotation R and translation t
    float R[9], a, bq, t2;
    const float v[3] = {rot_vec[3 * b], rot_vec[3 * b + 1], rot_vec[3 * b + 2]};
    rodrigues(v, R, &a, &bq, &t2);
    const float t[3] = {trans_vec[3 * b], trans_vec[3 * b + 1], trans_vec[3 * b + 2]};
    float G[9], dt[3];
    if (opt_cam_pose) {
#pragma unroll
        for (int i = 0; i < 9; ++i) G[i] = gR[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) dt[i] = gt[i];
    } else {        // Rc = R^T, tc = -R^T t
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int i = 0; i < 3; ++i) G[3 * j + i] = gR[3 * i + j] - gt[i] * t[j];
            dt[j] = -(R[3 * j] * gt[0] + R[3 * j + 1] * gt[1] + R[3 * j + 2] * gt[2]);
        }
    }
    // R = I + a K + b K^2:  dL/dv_k = a'_k <G,K> + a <G,E_k> + b'_k <G,K^2> + b <G K^T + K^T G, E_k>
    const float x = v[0], y = v[1], z = v[2];
    const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    float K2[9], M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
            // (G K^T)_ij = sum_m G_im K_jm ; (K^T G)_ij = sum_m K_mi G_mj
            M[3 * i + j] = (G[3 * i] * K[3 * j] + G[3 * i + 1] * K[3 * j + 1] + G[3 * i + 2] * K[3 * j + 2])
                         + (K[i] * G[j] + K[3 + i] * G[3 + j] + K[6 + i] * G[6 + j]);
        }
    float gk = 0.f, gk2 = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) { gk += G[i] * K[i]; gk2 += G[i] * K2[i]; }
    const bool small = t2 < 1e-8f;
    const float th = sqrtf(fmaxf(t2, 1e-24f));
    const float da = small ? -1.f / 3.f : (th * cosf(th) - sinf(th)) / (th * t2);                      // (da/dv_k) / v_k
    const float db = small ? -1.f / 12.f : (th * sinf(th) - 2.f * (1.f - cosf(th))) / (t2 * t2);       // (db/dv_k) / v_k
    const float e[3] = {G[7] - G[5], G[2] - G[6], G[3] - G[1]};
    const float m[3] = {M[7] - M[5], M[2] - M[6], M[3] - M[1]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (d_rot_vec) d_rot_vec[3 * b + k] = v[k] * (da * gk + db * gk2) + a * e[k] + bq * m[k];
        if (d_trans_vec) d_trans_vec[3 * b + k] = dt[k];
    }
}

// metrics (B,4) of one iteration: [PSNR of the foreground MSE, mean |depth - depth0| over the lidar pixels, rotation error, translation error]
__global__ void __launch_bounds__(64) metric_row_kernel(const float* __restrict__ loss_out /* (B,4): mse_fg in column 3 */,
                                                        const float* __restrict__ d_vec, float* __restrict__ depth0, int n_lidar, int first,
                                                        const float* __restrict__ cam2opt, const float* __restrict__ gt_R,
                                                        const float* __restrict__ gt_T, int opt_cam_pose, float* __restrict__ row,
                                                        const int32_t* __restrict__ lidar_count) {
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    float s = 0.f;
    // every object averages ITS OWN depth pixels: the first lidar_count[b] of the n_lidar columns (the rest is padding)
    const int cnt = lidar_count ? min(max(lidar_count[b], 0), n_lidar) : n_lidar;
    for (int i = lane; i < cnt; i += 64) {
        const float d = d_vec[b * n_lidar + i];
        if (first) depth0[b * n_lidar + i] = d;
        s += fabsf(d - (first ? d : depth0[b * n_lidar + i]));
    }
    s = wave_sum(s);
    if (lane != 0) return;
    const float* c = cam2opt + b * 12;
    float pR[9], pt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) pR[3 * i + j] = opt_cam_pose ? c[4 * i + j] : c[4 * j + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pt[i] = opt_cam_pose ? c[4 * i + 3] : -(pR[3 * i] * c[3] + pR[3 * i + 1] * c[7] + pR[3 * i + 2] * c[11]);
    // geodesic angle (src/utils.py:713-722): trace(pR gR^T) = sum_ij pR_ij gR_ij
    float tr = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) tr += pR[i] * gt_R[b * 9 + i];
    tr = fminf(fmaxf(tr, -1.f), 3.f);
    const float ang = acosf(fminf(fmaxf((tr - 1.f) * 0.5f, -1.f), 1.f));
    const float ex = pt[0] - gt_T[b * 3], ey = pt[1] - gt_T[b * 3 + 1], ez = pt[2] - gt_T[b * 3 + 2];
    row[b * 4 + 0] = -10.f * log10f(loss_out[b * 4 + 3]);
    row[b * 4 + 1] = s / ((float)cnt + 1e-8f);           // np.sum(depth_errs) / (len(gt_depth_vec) + 1e-8), src/optimizer_nuscenes.py:1741
    row[b * 4 + 2] = ang;
    row[b * 4 + 3] = sqrtf(ex * ex + ey * ey + ez * ez);
}

// torch.optim.AdamW (single-tensor arithmetic, amsgrad off) over up to 4 parameter groups in one launch
struct AdamGroups {
    float* p[4]; const float* g[4]; float* m[4]; float* v[4];
    long long n[4];
    float decay[4];        // 1 - lr * weight_decay
    float step_size[4];    // lr / (1 - beta1^step)
    int count;
};
// one element of torch.optim.AdamW's update (amsgrad off; torch/optim/adam.py's arithmetic, scalars prepared on the host in double)
__device__ __forceinline__ void adamw_one(float& p, float& m, float& v, float g, float decay, float step_size, float beta1, float beta2, float eps,
                                          float bias2_sqrt) {
    p *= decay;
    m = m + (g - m) * (1.f - beta1);                  // lerp
    v = v * beta2 + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p -= step_size * (m / denom);
}
__global__ void __launch_bounds__(256) adamw_kernel(AdamGroups a, float beta1, float beta2, float eps, float bias2_sqrt) {
    const int gi = blockIdx.y;
    if (gi >= a.count) return;
    const float step_size = a.step_size[gi], decay = a.decay[gi];
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < a.n[gi]; i += (long long)gridDim.x * 256) {
        float p = a.p[gi][i], m = a.m[gi][i], v = a.v[gi][i];
        adamw_one(p, m, v, a.g[gi][i], decay, step_size, beta1, beta2, eps, bias2_sqrt);
        a.p[gi][i] = p; a.m[gi][i] = m; a.v[gi][i] = v;
    }
}
// The same update for ANY number of tensors in one launch (the training step: 28 decoder tensors + latent layers + the two code tables):
// a table in device memory names every tensor once (the addresses of parameters, gradient views and moments do not change between
// steps), blockIdx.y picks the tensor, its parameter group picks the rate.
struct AdamEntry { float* p; const float* g; float* m; float* v; long long n; long long group; };
struct AdamScalars { float decay[4], step_size[4]; };
__global__ void __launch_bounds__(256) adamw_table_kernel(const AdamEntry* __restrict__ table, AdamScalars sc, float beta1, float beta2, float eps,
                                                          float bias2_sqrt) {
    const AdamEntry e = table[blockIdx.y];
    const float step_size = sc.step_size[e.group], decay = sc.decay[e.group];
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < e.n; i += (long long)gridDim.x * 256) {
        float p = e.p[i], m = e.m[i], v = e.v[i];
        adamw_one(p, m, v, e.g[i], decay, step_size, beta1, beta2, eps, bias2_sqrt);
        e.p[i] = p; e.m[i] = m; e.v[i] = v;
    }
}

// ---- the per-object latent layers (src/model_supnerf.py:253,261: z_j = ReLU(Lin_j(code)), the code = the shape code for the shape blocks, the
// texture code for the texture blocks) and the bias each z_j folds into (snr_render_args::latent_bias: b_next_j + W_next_j z_j), one launch.
// w_lat (512, n_lat*256): row block 0 multiplies the shape code, row block 1 the texture code, column block j = layer j (transposed
// nn.Linear weights, zero where a layer does not read that code); w_nxt (n_lat*256, n_lat*256) block diagonal, transposed likewise.
// One workgroup per (object, latent layer), a thread per output unit and quarter of its sum: every weight row is read with consecutive lanes on
// consecutive floats.
__global__ void __launch_bounds__(1024) latent_fwd_kernel(const float* __restrict__ sc, const float* __restrict__ tc, const float* __restrict__ w_lat,
                                                          const float* __restrict__ b_lat, const float* __restrict__ w_nxt, const float* __restrict__ b_nxt,
                                                          int sb, int n_lat, float* __restrict__ z, float* __restrict__ lb) {
    // 1024 threads: output unit t = tid & 255, the 256 terms of its sum in four quarters kq = tid >> 8 that meet in LDS (the kernel is a
    // chain of load latencies: four times the threads, a quarter of the trips)
    __shared__ float code[256], zs[256], quarter[4][256];
    const int b = blockIdx.x, j = blockIdx.y, t = threadIdx.x & 255, kq = threadIdx.x >> 8;
    const long long ld = (long long)n_lat * 256;
    if (kq == 0) code[t] = (j < sb ? sc : tc)[(long long)b * 256 + t];
    __syncthreads();
    auto dot64 = [&](const float* v, const float* w) {       // sum over k = 64 kq .. 64 kq + 63 of v[k] * w[k * ld]
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
        for (int k = 64 * kq; k < 64 * kq + 64; k += 4) {
            a0 = fmaf(v[k], w[(long long)k * ld], a0);           a1 = fmaf(v[k + 1], w[(long long)(k + 1) * ld], a1);
            a2 = fmaf(v[k + 2], w[(long long)(k + 2) * ld], a2); a3 = fmaf(v[k + 3], w[(long long)(k + 3) * ld], a3);
        }
        return (a0 + a1) + (a2 + a3);
    };
    quarter[kq][t] = dot64(code, w_lat + (long long)(j < sb ? 0 : 256) * ld + j * 256 + t);
    __syncthreads();
    if (kq == 0) {
        const float zv = fmaxf(((quarter[0][t] + quarter[1][t]) + (quarter[2][t] + quarter[3][t])) + b_lat[j * 256 + t], 0.f);
        z[((long long)b * n_lat + j) * 256 + t] = zv;
        zs[t] = zv;
    }
    if (!lb) return;
    __syncthreads();
    quarter[kq][t] = dot64(zs, w_nxt + (long long)(j * 256) * ld + j * 256 + t);
    __syncthreads();
    if (kq == 0) lb[((long long)b * n_lat + j) * 256 + t] = ((quarter[0][t] + quarter[1][t]) + (quarter[2][t] + quarter[3][t])) + b_nxt[j * 256 + t];
}
// backward to the codes: d code[k] = sum over the layers that read this code and their units t of (d z_j[t] where z_j[t] > 0) * w_lat[k][j*256 + t].
// One wave per code element (grid: objects x 2 codes x 64 workgroups of 4 waves), its lanes along the row of w_lat; fixed summation order.
__global__ void __launch_bounds__(256) latent_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ w_lat,
                                                         int sb, int n_lat, float* __restrict__ d_sc, float* __restrict__ d_tc) {
    const int b = blockIdx.x, which = blockIdx.y, lane = threadIdx.x & 63;
    const int k = blockIdx.z * 4 + (threadIdx.x >> 6);
    const int j0 = which ? sb : 0, j1 = which ? n_lat : sb;
    float* out = which ? d_tc : d_sc;
    if (!out) return;
    const long long ld = (long long)n_lat * 256;
    const float* w = w_lat + (long long)(which * 256 + k) * ld;
    const float* g = dz + (long long)b * ld;
    const float* zz = z + (long long)b * ld;
    float acc = 0.f;
    for (int c = j0 * 256 + lane; c < j1 * 256; c += 64) acc = fmaf(zz[c] > 0.f ? g[c] : 0.f, w[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[(long long)b * 256 + k] = acc;
}

}  // namespace snr

using namespace snr;

extern "C" {

int snr_pose_rays_fwd(const float* rot_vec, const float* trans_vec, const float* cam_dirs, const float* half_diag, const float* jitter,
                      int64_t n_objects, int64_t rays_per_obj, int n_samples, int opt_cam_pose,
                      float* cam2opt, float* rays_o, float* viewdir, float* z_vals, void* stream) {
    if (n_objects == 0) return SNR_OK;
    if (!rot_vec || !trans_vec || !cam_dirs || !rays_o || !viewdir) return SNR_E_ARG;
    if (n_objects < 0 || rays_per_obj < 0) return SNR_E_ARG;
    if (z_vals && (!half_diag || n_samples < 1 || n_samples > 256)) return SNR_E_ARG;
    const unsigned chunks = (unsigned)((rays_per_obj + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    pose_rays_fwd_kernel<false><<<dim3((unsigned)n_objects, chunks ? chunks : 1), 256, 0, (hipStream_t)stream>>>(
        rot_vec, trans_vec, cam_dirs, half_diag, jitter, rays_per_obj, n_samples, opt_cam_pose, cam2opt, rays_o, viewdir, z_vals);
    return snr_check_launch_();
}

int snr_cam_rays_fwd(const float* c2w, const float* cam_dirs, const float* half_diag, const float* jitter, int64_t n_objects,
                     int64_t rays_per_obj, int n_samples, float* rays_o, float* viewdir, float* z_vals, void* stream) {
    if (n_objects == 0) return SNR_OK;
    if (!c2w || !cam_dirs || !rays_o || !viewdir) return SNR_E_ARG;
    if (n_objects < 0 || rays_per_obj < 0) return SNR_E_ARG;
    if (z_vals && (!half_diag || n_samples < 1 || n_samples > 256)) return SNR_E_ARG;
    const unsigned chunks = (unsigned)((rays_per_obj + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    pose_rays_fwd_kernel<true><<<dim3((unsigned)n_objects, chunks ? chunks : 1), 256, 0, (hipStream_t)stream>>>(
        c2w, nullptr, cam_dirs, half_diag, jitter, rays_per_obj, n_samples, 1, nullptr, rays_o, viewdir, z_vals);
    return snr_check_launch_();
}

int snr_cam_rays_bwd(const float* c2w, const float* cam_dirs, int64_t n_objects, int64_t rays_per_obj, const float* d_rays_o,
                     const float* d_viewdir, float* d_c2w, void* stream) {
    if (n_objects == 0) return SNR_OK;
    if (!c2w || !cam_dirs || !d_c2w) return SNR_E_ARG;
    if (n_objects < 0 || rays_per_obj < 0) return SNR_E_ARG;
    pose_rays_bwd_kernel<true><<<(unsigned)n_objects, rays_per_obj >= 2048 ? 1024 : 256, 0, (hipStream_t)stream>>>(c2w, nullptr, cam_dirs, rays_per_obj, 1, d_rays_o, d_viewdir,
                                                                                     nullptr, d_c2w, nullptr);
    return snr_check_launch_();
}

int snr_pose_rays_bwd(const float* rot_vec, const float* trans_vec, const float* cam_dirs, int64_t n_objects, int64_t rays_per_obj,
                      int opt_cam_pose, const float* d_rays_o, const float* d_viewdir, const float* d_cam2opt,
                      float* d_rot_vec, float* d_trans_vec, void* stream) {
    if (n_objects == 0) return SNR_OK;
    if (!rot_vec || !trans_vec || !cam_dirs || (!d_rot_vec && !d_trans_vec)) return SNR_E_ARG;
    if (n_objects < 0 || rays_per_obj < 0) return SNR_E_ARG;
    pose_rays_bwd_kernel<false><<<(unsigned)n_objects, rays_per_obj >= 2048 ? 1024 : 256, 0, (hipStream_t)stream>>>(rot_vec, trans_vec, cam_dirs, rays_per_obj, opt_cam_pose,
                                                                                      d_rays_o, d_viewdir, d_cam2opt, d_rot_vec, d_trans_vec);
    return snr_check_launch_();
}

int snr_metric_row(const float* loss_out, const float* depth_pred, float* depth0, int n_lidar, int first, const float* cam2opt,
                   const float* gt_R, const float* gt_T, int64_t n_objects, int opt_cam_pose, float* row, const int32_t* lidar_count,
                   void* stream) {
    if (n_objects == 0) return SNR_OK;
    if (!loss_out || !cam2opt || !gt_R || !gt_T || !row || n_objects < 0 || n_lidar < 0) return SNR_E_ARG;
    if (n_lidar > 0 && (!depth_pred || !depth0)) return SNR_E_ARG;
    metric_row_kernel<<<(unsigned)n_objects, 64, 0, (hipStream_t)stream>>>(loss_out, depth_pred, depth0, n_lidar, first, cam2opt, gt_R, gt_T,
                                                                           opt_cam_pose, row, lidar_count);
    return snr_check_launch_();
}

int snr_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                   const float* lr, int n_groups, int64_t step, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (n_groups == 0) return SNR_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || n_groups < 0 || n_groups > 4 || step < 1) return SNR_E_ARG;
    AdamGroups a;
    long long nmax = 0;
    // bias corrections and the per-group scalars in double on the host, like torch's python-scalar path (torch/optim/adam.py)
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
    for (int i = 0; i < 4; ++i) {
        const bool on = i < n_groups;
        if (on && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0)) return SNR_E_ARG;
        a.p[i] = on ? params[i] : nullptr; a.g[i] = on ? grads[i] : nullptr; a.m[i] = on ? exp_avg[i] : nullptr; a.v[i] = on ? exp_avg_sq[i] : nullptr;
        a.n[i] = on ? numel[i] : 0;
        a.decay[i] = on ? (float)(1.0 - (double)lr[i] * (double)weight_decay) : 1.f;
        a.step_size[i] = on ? (float)((double)lr[i] / b1) : 0.f;
        if (a.n[i] > nmax) nmax = a.n[i];
    }
    a.count = n_groups;
    if (nmax == 0) return SNR_OK;
    long long gx = (nmax + 255) / 256; if (gx > 1024) gx = 1024;
    adamw_kernel<<<dim3((unsigned)gx, (unsigned)n_groups), 256, 0, (hipStream_t)stream>>>(a, beta1, beta2, eps, (float)sqrt(b2));
    return snr_check_launch_();
}

int snr_adamw_table_step(const void* table, int n_tensors, int64_t max_numel, const float* lr, int n_groups, int64_t step, float beta1, float beta2,
                         float eps, float weight_decay, void* stream) {
    if (n_tensors == 0) return SNR_OK;
    if (!table || !lr || n_tensors < 0 || n_tensors > 65535 || max_numel < 0 || n_groups < 1 || n_groups > 4 || step < 1) return SNR_E_ARG;
    if (((uintptr_t)table & 7) != 0) return SNR_E_ARG;
    static_assert(sizeof(AdamEntry) == 48, "table entry = 6 x 8 bytes");
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
    AdamScalars sc;
    for (int i = 0; i < 4; ++i) {
        const bool on = i < n_groups;
        sc.decay[i] = on ? (float)(1.0 - (double)lr[i] * (double)weight_decay) : 1.f;
        sc.step_size[i] = on ? (float)((double)lr[i] / b1) : 0.f;
    }
    if (max_numel == 0) return SNR_OK;
    long long gx = (max_numel + 1023) / 1024; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
    adamw_table_kernel<<<dim3((unsigned)gx, (unsigned)n_tensors), 256, 0, (hipStream_t)stream>>>((const AdamEntry*)table, sc, beta1, beta2, eps,
                                                                                                   (float)sqrt(b2));
    return snr_check_launch_();
}

int snr_latent_fwd(const float* shapecode, const float* texturecode, const float* w_lat, const float* b_lat, const float* w_nxt, const float* b_nxt,
                   int64_t n_objects, int shape_blocks, int texture_blocks, float* z, float* latent_bias, void* stream) {
    const int n_lat = shape_blocks + texture_blocks;
    if (n_objects == 0 || n_lat == 0) return SNR_OK;
    if (n_objects < 0 || n_objects > 0x7fffffff || shape_blocks < 0 || texture_blocks < 0 || n_lat > 65535) return SNR_E_ARG;
    if (!w_lat || !b_lat || !z || (shape_blocks && !shapecode) || (texture_blocks && !texturecode) || (latent_bias && (!w_nxt || !b_nxt))) return SNR_E_ARG;
    latent_fwd_kernel<<<dim3((unsigned)n_objects, (unsigned)n_lat), 1024, 0, (hipStream_t)stream>>>(shapecode, texturecode, w_lat, b_lat, w_nxt, b_nxt,
                                                                                                shape_blocks, n_lat, z, latent_bias);
    return snr_check_launch_();
}

int snr_latent_bwd(const float* d_z, const float* z, const float* w_lat, int64_t n_objects, int shape_blocks, int texture_blocks,
                   float* d_shapecode, float* d_texturecode, void* stream) {
    const int n_lat = shape_blocks + texture_blocks;
    if (n_objects == 0 || n_lat == 0) return SNR_OK;        /* (no latent layer: the codes receive no gradient; the caller zero-fills) */
    if (n_objects < 0 || n_objects > 0x7fffffff || shape_blocks < 0 || texture_blocks < 0) return SNR_E_ARG;
    if (!d_z || !z || !w_lat) return SNR_E_ARG;
    latent_bwd_kernel<<<dim3((unsigned)n_objects, 2, 64), 256, 0, (hipStream_t)stream>>>(d_z, z, w_lat, shape_blocks, n_lat, d_shapecode, d_texturecode);
    return snr_check_launch_();
}

}  // extern "C"
