/*
 * supnerf_hip.h -- C ABI of libsupnerf_hip.so: the MI355X (gfx950) implementation of
 * SUP-NeRF's volumetric rendering hot path.
 *
 * The reference (abhi1kumar/SUP-NeRF) has no FFI/plugin layer: its callers bind ordinary
 * Python functions by name.  The entry points below are therefore
 * what a ctypes/cffi stub inside the reference's src/utils.py, src/renderer.py and
 * src/model_supnerf.py would bind to replace the bodies of those functions; each one cites
 * the reference lines it replaces (paths relative to the reference repo).  INTEGRATION.md
 * shows the binding.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch allocation),
 *     fp32, contiguous, 16-byte aligned where noted; the library never allocates or frees
 *     caller-visible memory and never synchronises with the host;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - returns 0 on success or a negative SNR_E_* code; nothing is thrown across the ABI;
 *   - re-entrant and thread-safe: no global mutable state.
 *
 * Layout vocabulary:  N rays, S samples per ray, P = N*S sample points, B objects
 * (object-major: ray r belongs to object r / (N/B), src/model_supnerf.py:246-249),
 * W = 256 hidden width / latent width, NLAT = shape_blocks + texture_blocks latent terms.
 */
#ifndef SUPNERF_HIP_H
#define SUPNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNR_ABI_VERSION 8

enum {
    SNR_OK = 0,
    SNR_E_ARG = -1,        /* null pointer / bad size / unsupported hyper-parameter */
    SNR_E_SHAPE = -2,      /* sizes inconsistent with each other */
    SNR_E_WORKSPACE = -3,  /* workspace too small */
    SNR_E_LAUNCH = -4,     /* hipLaunchKernel failed (see snr_last_hip_error) */
    SNR_E_UNSUPPORTED = -5
};

/* how per-sample depths are laid out.
 * SNR_Z_BOX: no depth table at all -- the kernel derives every ray's depths itself, the way family B does
 * (NeRFRenderer.prepare_sampled_rays + sample_from_ray, src/renderer.py:27-41,91-115):
 *   o_n = rays_o / z_scale[obj]                           (`rays_o / (obj_diag / 2)`, :103)
 *   slab test of (o_n, rays_d) against the box +-box_half[obj] (ray_box_intersection_tensor, src/utils.py:283-327; NaN-propagating
 *   min / max like torch's), near = far = -1 for rays that miss (:106-108)
 *   u_s = s / S + jitter * (1 / S),  t_s = near (1 - u_s) + far u_s          (:33-41; S a power of two)
 *   p = o_n + t d (xyz_div is NOT applied), composite depth |p - o_n| z_scale with SNR_METRIC_Z (:114)
 * `t_vals` then is the (N,S) jitter in [0,1) (the reference's rand_like draw), or NULL: the kernel draws it itself with
 * Philox4x32-10 (see rng_* below).  The backward kernel returns the gradient THROUGH the bounds to rays_o / rays_d like the
 * reference's autograd (maximum / minimum split ties evenly, as torch does) unless SNR_BOX_DETACH is set. */
enum { SNR_Z_SHARED = 0 /* (S,) */, SNR_Z_PER_OBJECT = 1 /* (B,S) */, SNR_Z_PER_RAY = 2 /* (N,S) */, SNR_Z_BOX = 3 /* none: box bounds */ };

/* arithmetic of the decoder GEMMs.  SNR_FP32: v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fmaf chain.
 * SNR_BF16X3: every fp32 operand split into two 16-bit pieces hi + lo, three 16-bit MFMAs per product, fp32 accumulate (5x less
 * matrix time).  The FORWARD launches carry fp16 pieces (22 bits per operand, relative error ~2^-22 per product; activations are
 * clamped to +-65504, weights likewise at packing time), the BACKWARD launches and snr_weight_grad bf16 pieces (~2^-17: gradients need
 * the exponent range).  Needs shape_blocks + texture_blocks <= 4 and whole 32-point tiles per object;
 * snr_precision_supported() tells. */
enum { SNR_FP32 = 0, SNR_BF16X3 = 1 };
int snr_precision_supported(int precision, int shape_blocks, int texture_blocks, int64_t points_per_obj);

/* flags for the render / composite entry points */
enum {
    SNR_WHITE_BKGD = 1,    /* rgb += 1 - sum(w)            (src/renderer.py:60-63,374-377) */
    SNR_METRIC_Z   = 2,    /* composite depth = |t*d|*z_scale (src/renderer.py:114) instead of t */
    SNR_BOX_DETACH = 4     /* SNR_Z_BOX: the bounds carry no gradient (render_rays_v3 runs its slab test in numpy, src/renderer.py:425-432) */
};

int snr_abi_version(void);
/* text of the last HIP runtime error seen by this thread (for SNR_E_LAUNCH) */
const char* snr_last_hip_error(void);

/* ------------------------------------------------------------------------------------
 * Decoder weights.  The decoder is src/model_supnerf.py:184-199 (== model_codenerf.py:22-37):
 * width 256, latent 256, num_xyz_freq 10, num_dir_freq 4 (every shipped config).
 * `tensors` is a HOST array of 2*(shape_blocks+texture_blocks+6) DEVICE pointers holding the
 * per-point layers in this order, weight then bias, nn.Linear layout (out,in):
 *   encoding_xyz.0, shape_layer_1.0 .. shape_layer_SB.0, encoding_shape, sigma.0,
 *   encoding_viewdir.0, texture_layer_1.0 .. texture_layer_TB.0, rgb.0, rgb.2
 * (the *_latent_layer_* tensors are per-object work and stay with the caller).
 * The packed buffer holds the k-chunked forward stream, the transposed stream for the
 * backward pass and the small vectors; its size in bytes is snr_packed_bytes().
 * ---------------------------------------------------------------------------------- */
size_t snr_packed_bytes(int shape_blocks, int texture_blocks);
int snr_pack_weights(const float* const* tensors, int n_tensors, int shape_blocks, int texture_blocks,
                     float* packed, void* stream);

/* ------------------------------------------------------------------------------------
 * Decoder on explicit sample points: replaces SUPNeRF.forward / CodeNeRF.forward
 * (src/model_supnerf.py:241-269, src/model_codenerf.py:39-63).
 *   xyz, viewdir : (P,3)   object-frame points and unit directions
 *   latent       : (B,NLAT,256) z_j = ReLU(Lin_j(code)), the terms added before each block
 *   points_per_obj = P / B
 *   sigmas (P), rgbs (P,3) outputs (softplus density, raw linear colour)
 *   relu_masks   : NULL, or snr_mask_bytes(P,...) bytes that the backward pass needs
 *   activations  : NULL, or (shape_blocks+texture_blocks+4, P, 256) floats: training mode, receives the INPUT of every
 *                  MFMA layer after the first (slot l = input of layer l+1; the last slot = input of rgb.2, 128 columns
 *                  used).  Needs relu_masks; both precisions.  Together with `layer_grads` of snr_decoder_bwd and the encodings of
 *                  snr_pe_points they are the X operands of the weight-gradient products dW_l = G_l^T X_l of snr_weight_grad (the weight
 *                  half of the backward of src/trainer_unified_nuscenes.py:334).
 * ---------------------------------------------------------------------------------- */
size_t snr_mask_bytes(int64_t n_points, int shape_blocks, int texture_blocks);
int snr_decoder_fwd(const float* xyz, const float* viewdir, const float* latent, const float* packed,
                    int64_t n_points, int64_t points_per_obj, int shape_blocks, int texture_blocks,
                    float* sigmas, float* rgbs, void* relu_masks, float* activations, int precision, void* stream);
/* gradients wrt latent (B,NLAT,256), xyz (P,3), viewdir (P,3) [each nullable] given d_sigmas (P) and d_rgbs (P,3).
 * layer_grads: NULL, or (shape_blocks+texture_blocks+4, P, 256) floats receiving the gradient wrt the pre-activation of
 * every MFMA layer (slot l = layer l; rgb.0 uses 128 columns), both precisions.  workspace: snr_decoder_bwd_ws_bytes(). */
size_t snr_decoder_bwd_ws_bytes(int64_t n_points, int64_t points_per_obj, int shape_blocks, int texture_blocks);
int snr_decoder_bwd(const float* xyz, const float* viewdir, const float* latent, const float* packed,
                    const void* relu_masks, const float* sigmas, const float* d_sigmas, const float* d_rgbs,
                    int64_t n_points, int64_t points_per_obj, int shape_blocks, int texture_blocks,
                    float* d_latent, float* d_xyz, float* d_viewdir, float* layer_grads,
                    void* workspace, size_t ws_bytes, int precision, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused render: sample points on rays -> frame transform -> PE -> decoder -> composite.
 * Replaces the body of render_rays_v2 / render_rays_specified / render_rays /
 * render_full_img after ray generation (src/utils.py:468-500, 523-549, 400-431, 566-600) and
 * of NeRFRenderer.render_rays / render_rays_v3 after the box test
 * (src/renderer.py:108-114,155-166,433-468).
 *   rays_o, rays_d : (N,3) origin and unit direction in the sampling frame
 *   t_vals         : depths along the ray, layout z_mode
 *   xyz_div        : (B,) points are DIVIDED by this (family A: obj_diag, src/utils.py:472)
 *   xyz_mul        : scalar multiplier applied next (render_rays_v3 adjust_scale, src/renderer.py:441)
 *   frame          : 9 floats, row-major 3x3 applied to points and directions afterwards
 *                    (sym flip, kitti2nusc, shapenet_obj_cood: src/utils.py:475-495)
 *   z_scale        : (B,) metric scale for SNR_METRIC_Z (family B: obj_diag/2), else unused
 *   rgb (N,3), depth (N), acc_trans (N) outputs; sigmas (P) / rgbs (P,3) optional per-point outputs
 * ---------------------------------------------------------------------------------- */
typedef struct snr_render_args {
    const float* rays_o;
    const float* rays_d;
    const float* t_vals;
    const float* xyz_div;
    const float* z_scale;
    const float* latent;
    const float* packed;
    float frame[9];
    float xyz_mul;
    int32_t z_mode;
    int32_t flags;
    int64_t n_rays;
    int64_t rays_per_obj;
    int32_t n_samples;
    int32_t shape_blocks;
    int32_t texture_blocks;
    int32_t precision;      /* SNR_FP32 / SNR_BF16X3 */
    /* optional (may be null), forward only, (B, NLAT, 256): for every latent term z_j the bias the NEXT layer's accumulators start
     * from, b + W z_j (z_j is added after a ReLU, so it only ever reaches that layer through W z_j; the reference computes
     * `shape_layer_j(y + z_j)`, src/model_supnerf.py:253-263).  With it the split-bf16 forward drops the latent add and its vector
     * loads from every epilogue (-4 %); `latent` is still what the gradient d_latent of snr_render_bwd refers to. */
    const float* latent_bias;
    /* SNR_Z_BOX only: (B,3) half extents of every object's box in the o_n frame, (l, w, h) / diag (src/renderer.py:96-99) */
    const float* box_half;
    /* SNR_Z_BOX with t_vals == NULL: jitter of point i = ray * S + s is the uniform that torch.rand_like of an (N,S) tensor would
     * hold at i for the device generator state (rng_seed, rng_offset) when its kernel runs rng_threads threads (Philox4x32-10:
     * key = seed, counter = (offset / 4 + i / (4 rng_threads), subsequence i % rng_threads), word (i / rng_threads) % 4, value
     * (w + 1) 2^-32 folded to [0,1)).  rng_threads == 0: counter = (offset / 4, subsequence i), word 0. */
    uint64_t rng_seed;
    uint64_t rng_offset;
    uint64_t rng_threads;
} snr_render_args;

int snr_render_fwd(const snr_render_args* a, float* rgb, float* depth, float* acc_trans,
                   float* sigmas, float* rgbs, void* relu_masks, void* stream);
/* backward of the above: upstream d_rgb (N,3), d_depth (N), d_acc (N) [each nullable = zero]
 * -> d_latent (B,NLAT,256), d_rays_o (N,3), d_rays_d (N,3), d_t (layout of t_vals, SNR_Z_PER_RAY only)
 * [each nullable].  Needs sigmas/rgbs/relu_masks saved by the forward. */
size_t snr_render_bwd_ws_bytes(const snr_render_args* a);
int snr_render_bwd(const snr_render_args* a, const float* sigmas, const float* rgbs, const void* relu_masks,
                   const float* d_rgb, const float* d_depth, const float* d_acc,
                   float* d_latent, float* d_rays_o, float* d_rays_d, float* d_t,
                   void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-object scene pixels: replaces the per-pixel depth sort + scatter + volume_rendering3(white) of
 * OptimizerDemo.vis_scene (scripts/demo.py:555-565).  Every pixel carries n_per_pixel = Nb * S samples (the S samples of each of Nb
 * objects; depth -1 marks empty space and carries sigma 0); they are merged by depth and composited.  Samples of exactly equal depth
 * collapse like the reference's scatter: the last one in memory order survives.
 * run_length: S when each object's S samples are contiguous and ascending in depth (what vis_scene produces: stratified samples along
 * the object's ray, or all -1) -- the lists are then MERGED (binary searches, O(n log S Nb) per pixel) instead of rank-sorted (O(n^2));
 * the order is verified per pixel and a pixel with an unsorted list silently takes the rank sort, so the hint can never change a result.
 * 0 = no such structure.  Lists of 32, 64 or 128 samples with n_per_pixel <= 256 take two launches on the stream: a fast pass for pixels
 * without equal depths inside or across increasing lists (it marks the others in `rgb`) and the general kernel for the marked pixels.
 * sigmas, z_vals (P, n); rgbs (P, n, 3) -> rgb (P,3), depth (P) [nullable], acc_trans (P) [nullable].
 * flags: SNR_WHITE_BKGD.  n_per_pixel <= 1706 (LDS).
 * ---------------------------------------------------------------------------------- */
int snr_scene_composite_fwd(const float* sigmas, const float* rgbs, const float* z_vals, int64_t n_pixels, int n_per_pixel, int run_length,
                            int flags, float* rgb, float* depth, float* acc_trans, void* stream);

/* ------------------------------------------------------------------------------------
 * Alpha composite alone: replaces volume_rendering2 / volume_rendering_batch
 * (src/utils.py:202-233), NeRFRenderer.volume_render (src/renderer.py:43-65) and
 * volume_rendering3 (src/renderer.py:355-379).
 *   sigmas (N,S), rgbs (N,S,3), z_vals per z_mode (per-object: ray r uses row r / rays_per_obj)
 * ---------------------------------------------------------------------------------- */
int snr_composite_fwd(const float* sigmas, const float* rgbs, const float* z_vals, int z_mode, int flags,
                      int64_t n_rays, int64_t rays_per_obj, int n_samples,
                      float* rgb, float* depth, float* acc_trans, void* stream);
/* d_z (layout of z_vals) is produced only for SNR_Z_PER_RAY; pass NULL otherwise */
int snr_composite_bwd(const float* sigmas, const float* rgbs, const float* z_vals, int z_mode, int flags,
                      int64_t n_rays, int64_t rays_per_obj, int n_samples,
                      const float* d_rgb, const float* d_depth, const float* d_acc,
                      float* d_sigmas, float* d_rgbs, float* d_z, void* stream);

/* ------------------------------------------------------------------------------------
 * Sample encoding alone: ray packet -> sample points (+ optional positional encoding).
 * Replaces sample_from_rays + the in-place frame edits (src/utils.py:154-167,472-495), the
 * point/metric-depth part of NeRFRenderer.prepare_sampled_rays (src/renderer.py:111-114) and
 * PE (src/model_supnerf.py:155-161).  Arguments as snr_render_args; outputs
 *   xyz (N,S,3), viewdir (N,S,3), z_out (N,S) [nullable], pe_xyz (N,S,63) [nullable],
 *   pe_dir (N,27) [nullable; constant along S], hit (N) [nullable; SNR_Z_BOX: 1 where the ray meets its box, the
 *   `intersect` map of prepare_sampled_rays]
 * ---------------------------------------------------------------------------------- */
int snr_encode_fwd(const snr_render_args* a, float* xyz, float* viewdir, float* z_out,
                   float* pe_xyz, float* pe_dir, uint8_t* hit, void* stream);
/* The per-object latent layers of a frozen decoder in one launch (src/model_supnerf.py:253,261; model_codenerf.py:50,58):
 *   z[b][j] = ReLU(code_j[b] W_j^T + b_j), j < shape_blocks reads the shape code, the rest the texture code  -> z (B, n_lat, 256)
 *   latent_bias[b][j] = b_next_j + z[b][j] W_next_j^T   [nullable]: what snr_render_args::latent_bias takes
 * with the weights STACKED and TRANSPOSED once by the host: w_lat (512, n_lat*256) -- rows 0..255 multiply the shape code, rows 256..511
 * the texture code, column block j = layer j, zero where a layer does not read that code -- b_lat (n_lat*256), w_nxt (n_lat*256, n_lat*256)
 * block diagonal (block j = shape_layer_{j+1} / texture_layer_{..}.0.weight^T), b_nxt (n_lat*256).  shapecode, texturecode (B,256).
 * snr_latent_bwd: d_z (B, n_lat, 256) -> d_shapecode, d_texturecode (B,256) [each nullable] through the ReLU (z > 0) and W_j; a code no
 * layer reads gets zeros.  (Weight gradients of the latent layers: not here -- training mode keeps them on torch.) */
int snr_latent_fwd(const float* shapecode, const float* texturecode, const float* w_lat, const float* b_lat, const float* w_nxt, const float* b_nxt,
                   int64_t n_objects, int shape_blocks, int texture_blocks, float* z, float* latent_bias, void* stream);
int snr_latent_bwd(const float* d_z, const float* z, const float* w_lat, int64_t n_objects, int shape_blocks, int texture_blocks,
                   float* d_shapecode, float* d_texturecode, void* stream);
/* Positional encodings of explicit points (PE, src/model_supnerf.py:155-161) in the layout the training step's weight-gradient products
 * read: xyz, viewdir (P,3) -> out (P,96), 16-byte aligned: columns 0..62 = PE(xyz, 10 frequencies), 63 = 0, 64..90 = PE(viewdir, 4
 * frequencies), 91..95 = 0 -- the input of encoding_xyz and the direction features of encoding_viewdir (X of their dW = G^T X). */
int snr_pe_points(const float* xyz, const float* viewdir, int64_t n_points, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Loss / metric tail of one optimise iteration: replaces the three masked reductions the callers run right after the render
 * (src/optimizer_nuscenes.py:729-744 == src/optimizer_kitti.py:792-812; per object of a batch:
 * src/trainer_unified_nuscenes.py:133-140).  Rays are object-major, rays_per_obj per object, B = n_rays / rays_per_obj:
 *   a = |occ|, den = sum(a) + 1e-9
 *   loss_rgb = sum((rgb - rgb_tgt)^2 a) / den        loss_occ = sum(exp(-occ (0.5 - acc_trans)) a) / den
 *   loss     = loss_rgb + loss_occ_coef * loss_occ   mse_fg   = sum((rgb - rgb_tgt)^2 max(occ, 0)) / (sum(max(occ, 0)) + 1e-9)
 * rgb, rgb_tgt (N,3); acc, occ (N) -> out (B,4) = [loss, loss_rgb, loss_occ, mse_fg]  (PSNR = -10 log10 mse_fg, :743).
 * snr_loss_tail_bwd writes the gradient seeds of the render backward, d(sum_b upstream_b loss_b)/d rgb (N,3) and /d acc (N)
 * [each nullable]; upstream (B,) nullable = ones.  One launch each, no host synchronisation.
 * ---------------------------------------------------------------------------------- */
int snr_loss_tail_fwd(const float* rgb, const float* acc, const float* rgb_tgt, const float* occ, int64_t n_rays, int64_t rays_per_obj,
                      float loss_occ_coef, float* out, void* stream);
int snr_loss_tail_bwd(const float* rgb, const float* acc, const float* rgb_tgt, const float* occ, int64_t n_rays, int64_t rays_per_obj,
                      float loss_occ_coef, const float* upstream, float* d_rgb, float* d_acc, void* stream);

/* ------------------------------------------------------------------------------------
 * The rest of one iteration of the test-time optimisation loop (src/optimizer_nuscenes.py:674-783 == src/optimizer_kitti.py:731-866),
 * one launch each, B objects per launch (one workgroup per object), nothing synchronises with the host.
 *
 * snr_pose_rays_fwd: the optimised parameters rot_vec (B,3) [axis-angle; pytorch3d axis_angle_to_matrix in the reference, :666,:686] and
 *   trans_vec (B,3) -> camera-in-object pose cam2opt (B,3,4) [= the inverse of the object pose unless opt_cam_pose, :690-699], then what
 *   render_rays_v2 derives from it: rays_o / unit viewdir (B*n,3) of the object's pixels (get_rays, src/utils.py:107-135; cam_dirs (B,n,3) =
 *   [(px-cx)/fx, (py-cy)/fy, 1] is constant over the loop), near/far = |camera centre| -/+ half_diag (:468-469) and the stratified depth
 *   vector z_vals (B,S) of sample_from_rays (:159-164) with the given jitter (B,S) [nullable = 0].  cam2opt / z_vals nullable.
 * snr_pose_rays_bwd: d_rays_o, d_viewdir (B*n,3), d_cam2opt (B,3,4) [each nullable] -> d_rot_vec, d_trans_vec (B,3).  The depths are
 *   detached from the pose like the reference's .tolist().
 * ---------------------------------------------------------------------------------- */
int snr_pose_rays_fwd(const float* rot_vec, const float* trans_vec, const float* cam_dirs, const float* half_diag, const float* jitter,
                      int64_t n_objects, int64_t rays_per_obj, int n_samples, int opt_cam_pose,
                      float* cam2opt, float* rays_o, float* viewdir, float* z_vals, void* stream);
int snr_pose_rays_bwd(const float* rot_vec, const float* trans_vec, const float* cam_dirs, int64_t n_objects, int64_t rays_per_obj,
                      int opt_cam_pose, const float* d_rays_o, const float* d_viewdir, const float* d_cam2opt,
                      float* d_rot_vec, float* d_trans_vec, void* stream);
/* The same two launches for a caller that holds the camera pose itself -- what the public get_rays / render_rays_v2 do per call
 * (src/utils.py:107-135 get_rays, :468-469 sphere bounds, :159-164 sample_from_rays' depth vector): c2w (B,3,4) row-major [R | t]
 * -> rays_o / unit viewdir (B*n,3) and the stratified depths z_vals (B,S) [z_vals, half_diag, jitter nullable]; backward:
 * d_rays_o, d_viewdir (B*n,3) [each nullable] -> d_c2w (B,3,4).  The depths are detached from the pose like the reference's. */
int snr_cam_rays_fwd(const float* c2w, const float* cam_dirs, const float* half_diag, const float* jitter, int64_t n_objects,
                     int64_t rays_per_obj, int n_samples, float* rays_o, float* viewdir, float* z_vals, void* stream);
int snr_cam_rays_bwd(const float* c2w, const float* cam_dirs, int64_t n_objects, int64_t rays_per_obj, const float* d_rays_o,
                     const float* d_viewdir, float* d_c2w, void* stream);
/* The metric row the loop logs every iteration (:739-765): row (B,4) = [PSNR = -10 log10(loss_out[:,3]), depth error, rotation error
 * rot_dist(pred_R, gt_R) (src/utils.py:713-722), translation error |pred_t - gt_T|]; gt_R (B,3,3), gt_T (B,3) are the true OBJECT pose,
 * cam2opt the current camera-in-object pose.  Depth error of object b = sum_i |depth_pred[b,i] - depth0[b,i]| / (count_b + 1e-8) over its
 * first count_b = lidar_count[b] (int32, nullable = n_lidar for every object) of the n_lidar columns: with depth0 = the lidar
 * measurements and first = 0 this is log_eval_depth_v2 (src/optimizer_nuscenes.py:751-765,1736-1741); first != 0 stores
 * depth0 <- depth_pred instead (objects without a depth map: the change of rendered depth against the first iteration is logged). */
int snr_metric_row(const float* loss_out, const float* depth_pred, float* depth0, int n_lidar, int first, const float* cam2opt,
                   const float* gt_R, const float* gt_T, int64_t n_objects, int opt_cam_pose, float* row, const int32_t* lidar_count,
                   void* stream);
/* torch.optim.AdamW's update (amsgrad off) of up to 4 parameter groups in one launch (:1762-1769): HOST arrays of n_groups device
 * pointers / element counts / learning rates; step = 1 for the first update. */
int snr_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                   const float* lr, int n_groups, int64_t step, float beta1, float beta2, float eps, float weight_decay, void* stream);
/* The same update for any number of tensors in ONE launch: the training step's optimiser (src/trainer_unified_nuscenes.py:414-422, AdamW over
 * every decoder tensor and the two code tables).  table: DEVICE memory, n_tensors entries of six 64-bit words
 * {param*, grad*, exp_avg*, exp_avg_sq*, numel, group} (fp32 contiguous tensors; group < n_groups selects the learning rate lr[group],
 * a HOST array); max_numel = the largest numel in the table; step = 1 for the first update. */
int snr_adamw_table_step(const void* table, int n_tensors, int64_t max_numel, const float* lr, int n_groups, int64_t step, float beta1, float beta2,
                         float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------------------------
 * Weight gradient of one decoder layer (training mode): the weight half of loss_total.mean().backward()
 * (src/trainer_unified_nuscenes.py:334) for y = x W^T + b (nn.Linear, src/model_supnerf.py:184-199):
 *     dW (n_out, n_in; leading dimension ld_dw) = G^T X,   db (n_out) = column sums of G [nullable]
 * G (P, n_out; leading dimension ldg) = gradient wrt the layer's pre-activation = slot l of snr_decoder_bwd's layer_grads;
 * X (P, n_in; leading dimension ldx) = the layer's input = slot l-1 of snr_decoder_fwd's activations (or the positional encoding).
 * precision SNR_FP32: exact fp32 on the matrix cores; SNR_BF16X3: the split-bf16 products of the render fast path (operand error ~2^-17,
 * 5x less matrix time, HBM-bound); the narrow heads and the bias sums are always fp32.  Split over the points, per-slice partials summed
 * in slice order (deterministic, no atomics).
 * n_out, n_in <= 256.  n_out >= 32: n_out, n_in, ldg, ldx multiples of 4 and G, X 16-byte aligned; n_out <= 4 (density / colour head):
 * no alignment requirement.  workspace: snr_weight_grad_ws_bytes().
 * ---------------------------------------------------------------------------------- */
size_t snr_weight_grad_ws_bytes(int64_t n_points, int n_out, int n_in);
int snr_weight_grad(const float* G, int64_t ldg, int n_out, const float* X, int64_t ldx, int n_in, int64_t n_points,
                    float* dW, int64_t ld_dw, float* db, int precision, void* workspace, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUPNERF_HIP_H */
