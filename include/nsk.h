/*
 * nsk.h -- C-ABI of the MI355X-native NICE-SLAM render / map / track hot path ("neural slam kernels").
 *
 * This is the drop-in boundary.  The reference exposes no FFI layer: its boundary is the
 * C++ class surface of include/Renderer.h:11-13, include/Mapper.h:20-25, include/Tracker.h:11-15 and
 * include/models/NICE.h:6-7, all of which hand libtorch tensors to stock libtorch ops.  The entry points below
 * are what thin Renderer / NICE / Mapper / Tracker classes with those exact signatures marshal into
 * (nice-slam-cpp_amd/host/); every entry point cites the reference code it replaces.
 *
 * Conventions
 *   - plain C, no torch types.  `d_` pointers are device (HIP) pointers, `h_` pointers are host pointers.
 *   - the caller owns every buffer it passes; the context owns grids, decoder parameters, their gradients
 *     and the Adam moments.
 *   - every function returns 0 on success, <0 on error; nsk_last_error() gives a thread-local message.
 *     (The reference defines no error behaviour: libtorch c10::Error exceptions propagate out of main.)
 *   - one nsk_ctx = one GPU + one HIP stream.  Calls are asynchronous on that stream unless stated; a context
 *     is not thread-safe, different contexts may be driven from different threads.
 *   - levels / stages / decoders share ids: 0 coarse, 1 middle, 2 fine, 3 color (src/models/NICE.cpp:16-52).
 *   - all arithmetic is fp32 (the reference's dtype); tolerance contract: 1e-4 relative L2 on rendered
 *     depth / colour and on optimised grids / poses.
 */
#ifndef NSK_H
#define NSK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSK_COARSE 0
#define NSK_MIDDLE 1
#define NSK_FINE 2
#define NSK_COLOR 3

/* nsk_render_backward / nsk_map_step `flags` */
#define NSK_GRAD_GRIDS 1u     /* accumulate d loss / d grid features of the levels the stage reads         */
#define NSK_GRAD_DECODERS 2u  /* accumulate d loss / d decoder parameters of decoders marked trainable     */
#define NSK_GRAD_RAYS 4u      /* write d loss / d rays_o, d loss / d rays_d (Tracker, BA)                  */

/* Adam parameter groups, in the order of torch::optim::Adam's groups at src/Mapper.cpp:330 */
#define NSK_GROUP_DECODERS 0
#define NSK_GROUP_COARSE 1
#define NSK_GROUP_MIDDLE 2
#define NSK_GROUP_FINE 3
#define NSK_GROUP_COLOR 4
#define NSK_GROUP_CAMERA 5
#define NSK_NUM_GROUPS 6
#define NSK_MAX_POSE_FRAMES 32     /* frames of one nsk_pose_step_multi call (a mapping window; color_refine doubles mapping_window_size 5 -> 10) */

typedef struct nsk_ctx nsk_ctx;

const char* nsk_last_error(void);
int nsk_version(void);

/* ---- context ------------------------------------------------------------------------------------------- */
/* Replaces the 33 hard-coded torch::Device(torch::kCUDA,0) sites (e.g. src/Renderer.cpp:31,69,76).
 * hip_stream: a hipStream_t to launch on (NULL = the context creates its own non-blocking stream). */
int nsk_ctx_create(int device, void* hip_stream, nsk_ctx** out);
int nsk_ctx_destroy(nsk_ctx* ctx);
int nsk_sync(nsk_ctx* ctx);                         /* hipStreamSynchronize */
void* nsk_stream(nsk_ctx* ctx);                     /* the hipStream_t in use */

/* How the MLP decoders' FORWARD matrix products are evaluated (results agree within the 1e-4 contract; measured against the fp64
 * oracle all three are equally close, tests/test_gpu_parity.py::test_forward_bf16_split_mode_matches_oracle):
 *   2 = fp32 operands split into two fp16 pieces (x = h + l/2048: 22 significant bits), three v_mfma_f32_16x16x32_f16 per K=32
 *       block, fp32 accumulation (DEFAULT).  Operand range: |x| < 65504 (fp16) for activations, grid features and weights -- far
 *       above anything these decoders produce; larger values come out as inf / NaN in the rendering, never silently wrong;
 *   1 = three bf16 pieces (24 significant bits, the full fp32 exponent range), six v_mfma_f32_16x16x32_bf16 per block;
 *   0 = v_mfma_f32_16x16x4_f32, plain fp32.
 * Independent of the mode: the backward chains of the frozen decoders (without ray gradients) and of a trainable middle / colour
 * decoder run on two fp16 pieces of a per-sample power-of-two multiple of the upstream gradient (exact scaling: no range
 * restriction); frozen chains that carry ray gradients, the coarse decoder and a trainable fine decoder on the fp32 MFMA; the
 * weight-gradient panels on two bf16 pieces with fp32 sums.
 * Changing the mode rebuilds the forward images of the loaded decoders and invalidates captured graphs.
 * The library reads no environment variables: this call and nsk_set_render_opts are the only behaviour switches. */
int nsk_set_matmul_mode(nsk_ctx* ctx, int mode);
/* The same choice for the BACKWARD's gradient chains (g_h -> W^T, fc^T products of every decoder of the stage): 2 (default) = two fp16 pieces
 * on a per-sample power-of-two multiple of the upstream gradient; 0 = the fp32 MFMA (full-width operands: what the reference's fp32 autograd
 * multiplies, src/Mapper.cpp:443-444) for the frozen AND the trainable roles -- a measuring stick for the gradient error and the step time of
 * full-width arithmetic (tests/test_gpu_configs.py, bench.py extras), 1.5-2x slower.  The trainable decoder's weight-gradient panels keep two
 * bf16 pieces (16 significant bits, fp32 sums over the samples) in both modes. */
int nsk_set_backward_mode(nsk_ctx* ctx, int mode);

/* Order in which the decoder kernels of a step walk the rays' samples (results differ only by the order of floating-point sums in the
 * gradients): -1 = automatic (DEFAULT: cell-sorted for steps that scatter into the grids without ray gradients and have >= 14336 samples -- about
 * 300 rays x 48, the measured crossover on the reference's grids: below it the two sort launches cost more than the scatter saves -- or at
 * least four samples per cell of the finest level read, where ray order serialises the atomics of the many samples that share a cell),
 * 0 = ray order always, 1 = cell-sorted always.  Cell-sorted: k_sample also bins every sample by the grid cell it falls in and two small
 * launches build the permutation; tiles of 16 samples then share cells and the backward issues one atomic flush per cell run. */
int nsk_set_sort_mode(nsk_ctx* ctx, int mode);
/* Debug and experiment switches (never needed for correct results):
 *   "deterministic" 1: bit-reproducible gradients, to tell a real defect from the order sensitivity of floating-point atomics: ray order
 *                      (no cell sort), one backward launch per decoder in a fixed order, ONE workgroup each whose waves add their tiles'
 *                      contributions strictly one after the other (orders of magnitude slower; the trainable decoder's pose gradients are
 *                      not covered);
 *   "roctx" 1:         roctxRangePush/Pop around every launch group (names as in nsk_profile_end) for rocprofv3 --marker-trace;
 *   "frozen_cost" n:   relative cost of a frozen decoder's tile in the backward's workgroup split (0 = built-in value);
 *   "no_fused_median" 1: nsk_track_step computes the Tracker's median threshold in a launch of its own (composite, median, composite)
 *                      even where a fused form applies;
 *   "no_deferred_median" 1: with ray gradients and frozen decoders the threshold is found inside the compositing launch behind a grid
 *                      barrier (round 3's form) instead of by the backward launch's workgroups;
 *   "no_piggyback" 1:  a batch registered with nsk_map_prepare is sampled by launches of its own at the start of its step;
 *   "no_occ_role" 1 | 2: the forward's middle and fine decoders never | always as one workgroup role (default: where the split predicts
 *                      the shorter launch; same results either way). */
int nsk_set_tuning(nsk_ctx* ctx, const char* key, int value);

/* Scene bound [[x0,x1],[y0,y1],[z0,z1]]; the reference hard-codes it in five places
 * (src/main.cpp:33, src/Renderer.cpp:15, src/Mapper.cpp:29, src/Tracker.cpp:23, src/models/MLP.cpp:53-56). */
int nsk_set_bound(nsk_ctx* ctx, const float h_bound[6]);

/* Renderer::Renderer() constants (src/Renderer.cpp:5-15): N_samples 32, N_surface 16, lindisp false, perturb 0,
 * occupancy: 0 = the density branch the reference executes (src/Renderer.cpp:125 passes false, utils.h:155-157),
 * 1 = alpha = sigmoid(10 sigma).  n_samples + n_surface <= 64. */
int nsk_set_render_opts(nsk_ctx* ctx, int n_samples, int n_surface, int lindisp, float perturb, int occupancy,
                        uint64_t seed);

/* ---- feature grids: c10::Dict<string,Tensor> "grid_<level>" [1,C,Z,Y,X] fp32 (src/main.cpp:33-78) ------ */
/* h_czyx is the reference layout [C][Z][Y][X] (C must be 32); the device copy is voxel-major [Z][Y][X][C]
 * so that the 32 channels of a voxel are one 128-byte line (conversion happens here). */
int nsk_grid_upload(nsk_ctx* ctx, int level, const float* h_czyx, int C, int Z, int Y, int X);
int nsk_grid_download(nsk_ctx* ctx, int level, float* h_czyx);
int nsk_grid_grad_download(nsk_ctx* ctx, int level, float* h_czyx);
/* frustum feature selection (src/Mapper.cpp:254-290,333-350): h_mask_zyx[Z*Y*X] != 0 marks voxels that are
 * optimiser parameters; NULL = all voxels.  Gradients of unmarked voxels are discarded. */
int nsk_set_mask(nsk_ctx* ctx, int level, const uint8_t* h_mask_zyx);

/* Mapper::get_mask_from_c2w (src/Mapper.cpp:42-130, intended semantics): builds the frustum mask of `level` on the device from
 * a depth image (d_depth [H][W], device) and the current pose h_c2w (16 floats, row-major [4][4]) and installs it like
 * nsk_set_mask; h_mask_out [Z*Y*X] (host, may be NULL) receives a copy.  A voxel is kept if its centre projects inside the image
 * with 0 <= depth_along_-z <= sampled_depth + 0.5 (zero depths count as the maximum sampled depth) or lies within 0.5 m of the
 * camera centre; grid_coarse keeps every voxel. */
int nsk_frustum_mask(nsk_ctx* ctx, int level, const float* d_depth, int H, int W, float fx, float fy, float cx, float cy,
                     const float h_c2w[16], uint8_t* h_mask_out);

/* Mapper::keyframe_selection_overlap (src/Mapper.cpp:132-196; include/torchlib/utils.h:58-130): the rays of N pixels of the
 * current frame (device arrays, e.g. from nsk_rays_from_pixels) are sampled at n_samples depths in [0.8 depth, depth + 0.5] and
 * projected into each of K keyframes (h_c2w: K row-major 4x4 poses, host); h_percent[k] = fraction of the points inside keyframe
 * k's image (20-pixel edge, in front of the camera).  Ranking and truncation to the window stay with the caller (Mapper). */
int nsk_keyframe_overlap(nsk_ctx* ctx, int N, const float* d_rays_o, const float* d_rays_d, const float* d_gt_depth, int n_samples,
                         int H, int W, float fx, float fy, float cx, float cy, int K, const float* h_c2w, float* h_percent);

/* ---- decoders (src/models/MLP.cpp:3-49,104-138; src/models/GaussianFFT.cpp:3-8) -------------------------- */
/* packed parameter order (row-major [out,in] as torch::nn::Linear):
 *   middle/fine/color: B[3][93], pts_linear[0..4].{weight,bias}, fc[0..4].{weight,bias}, output_linear.{weight,bias}
 *   coarse:            pts_linear[0..4].{weight,bias}, output_linear.{weight,bias}
 * counts: coarse 6337, middle 15800, fine 20920, color 15899. */
size_t nsk_decoder_param_count(int which);
int nsk_decoder_upload(nsk_ctx* ctx, int which, const float* h_packed, size_t n);
int nsk_decoder_download(nsk_ctx* ctx, int which, float* h_packed, size_t n);
int nsk_decoder_grad_download(nsk_ctx* ctx, int which, float* h_packed, size_t n);
/* which decoders receive gradients / Adam updates (src/Mapper.cpp:292-301: fine if !fix_fine, color if !fix_color).  Set it
 * before the forward of the step: the forward of a trainable decoder stores its block outputs for the backward (the entry points
 * that contain both -- nsk_map_step, nsk_track_step, nsk_render_backward -- always do); a backward that finds them stale fails. */
int nsk_decoder_set_trainable(nsk_ctx* ctx, int which, int trainable);

/* ---- rendering ------------------------------------------------------------------------------------------ */
/* Renderer::render_batch_ray (src/Renderer.cpp:44-126) = z sampling + Renderer::eval_points (:19-42) +
 * NICE::forward (src/models/NICE.cpp:16-52) + raw2outputs_nerf_color (include/torchlib/utils.h:148-172).
 *   d_rays_o, d_rays_d [N][3]; d_gt_depth [N] or NULL (then N_surface = 0, :54-57);
 *   gt_depth_max: max over the WHOLE batch of gt_depth (:76,:93).  Pass < 0 to have it computed on the device
 *   from d_gt_depth; multi-GPU callers pass the global maximum so that sharding rays does not change results.
 *   outputs: d_rgb [N][3], d_depth [N], d_var [N], d_weights [N][S] or NULL (S = n_samples (+ n_surface)). */
int nsk_render_forward(nsk_ctx* ctx, int stage, int N, const float* d_rays_o, const float* d_rays_d,
                       const float* d_gt_depth, float gt_depth_max, float* d_rgb, float* d_depth, float* d_var,
                       float* d_weights);

/* Renderer::eval_points (src/Renderer.cpp:19-42): raw [M][4] = (rgb, occupancy) of M points, occupancy = 100
 * outside the bound. */
int nsk_eval_points(nsk_ctx* ctx, int stage, int M, const float* d_points, float* d_raw);

/* raw2outputs_nerf_color (include/torchlib/utils.h:148-172) on its own: d_raw [N][S][4] (rgb, sigma), d_z [N][S] sorted,
 * d_rays_d [N][3]; occupancy 0 = density branch (what src/Renderer.cpp:125 passes). */
int nsk_raw2outputs(nsk_ctx* ctx, int N, int S, const float* d_raw, const float* d_z, const float* d_rays_d, int occupancy,
                    float* d_rgb, float* d_depth, float* d_var, float* d_weights);

/* Backward of render_batch_ray given upstream gradients (what loss.backward() at src/Mapper.cpp:444 /
 * src/Tracker.cpp:84 is meant to do).  The forward is recomputed internally.
 *   d_g_rgb [N][3], d_g_depth [N], d_g_var [N] or NULL (depth_var detached).
 *   Gradients ACCUMULATE into the context-owned gradient slab (grids, trainable decoders) until nsk_adam_step
 *   or nsk_zero_grads; d_g_rays_o / d_g_rays_d [N][3] are overwritten (NSK_GRAD_RAYS). */
int nsk_render_backward(nsk_ctx* ctx, int stage, int N, const float* d_rays_o, const float* d_rays_d,
                        const float* d_gt_depth, float gt_depth_max, const float* d_g_rgb, const float* d_g_depth,
                        const float* d_g_var, unsigned flags, float* d_g_rays_o, float* d_g_rays_d);

/* One mapping iteration without the Adam step: render forward, Mapper loss (src/Mapper.cpp:435-442:
 * sum_{gt>0}|gt_d - d| + use_color * w_color * sum|gt_c - c|) and backward, fused so that the loss gradient
 * never leaves the GPU.  d_loss (device float, may be NULL) receives the loss; outputs d_rgb/d_depth/d_var may
 * be NULL. */
int nsk_map_step(nsk_ctx* ctx, int stage, int N, const float* d_rays_o, const float* d_rays_d,
                 const float* d_gt_depth, const float* d_gt_color, float gt_depth_max, float w_color, int use_color,
                 unsigned flags, float* d_loss, float* d_rgb, float* d_depth, float* d_var, float* d_g_rays_o,
                 float* d_g_rays_d);

/* One tracking iteration without the Adam step (src/Tracker.cpp:41-89): render, dynamic-outlier mask
 * |gt_d - d| < 10 median (:67-71), loss sum_mask |gt_d - d| / sqrt(var + 1e-10) + w_color sum_mask |gt_c - c|
 * (:75-82), backward onto the rays.  detach_var: treat depth_var as a constant.
 * With handle_dynamic and at most 1024 rays the median needs no launch of its own.  With NSK_GRAD_RAYS and no trainable decoder (the
 * Tracker as the reference runs it) the loss launch writes the residuals and every workgroup of the backward launch selects the
 * median and drops the rays that fail it; otherwise (and while the grid is at most one 4-ray workgroup per CU) the
 * median is found inside the loss launch: the residuals meet at a device-wide barrier whose wait is bounded; should it ever time
 * out, that step ran with an infinite threshold and the next nsk_sync returns the error. */
int nsk_track_step(nsk_ctx* ctx, int stage, int N, const float* d_rays_o, const float* d_rays_d,
                   const float* d_gt_depth, const float* d_gt_color, float gt_depth_max, float w_color, int use_color,
                   int handle_dynamic, int detach_var, unsigned flags, float* d_loss, float* d_g_rays_o,
                   float* d_g_rays_d);

/* Stand-alone losses with seed gradients (same formulas as above) for callers that drive
 * nsk_render_forward / nsk_render_backward themselves. */
int nsk_loss_map(nsk_ctx* ctx, int N, const float* d_depth, const float* d_rgb, const float* d_gt_depth,
                 const float* d_gt_color, float w_color, int use_color, float* d_g_depth, float* d_g_rgb,
                 float* d_loss);
int nsk_loss_track(nsk_ctx* ctx, int N, const float* d_depth, const float* d_rgb, const float* d_var,
                   const float* d_gt_depth, const float* d_gt_color, float w_color, int use_color, int handle_dynamic,
                   int detach_var, float* d_g_depth, float* d_g_rgb, float* d_g_var, float* d_loss);

/* ---- rays and pose (include/torchlib/utils.h:13-55,141-146,174-210; src/Mapper.cpp:416-427) ------------- */
/* raySampler's direction/origin part for given pixel indices (the reference draws them with torch::randint,
 * utils.h:32; streams cannot match, so indices are an input).  d_c2w: 12 floats, row-major [3][4].
 * mode bit0: as-written j_t=(i-cy)/fy without sign flip (D11); bit1: truncate intrinsics to int (D10);
 * 0 = intended OpenGL camera dirs=[(i-cx)/fx, -(j-cy)/fy, -1]. */
/* raySampler's pixel draw (utils.h:19-36: n indices, with replacement, uniform over the window [H0,H1) x [W0,W1)) and its
 * gather of the ground truth (utils.h:38-43) on the device.  The draw uses a counter-based hash of (seed, ray index) instead of
 * torch::randint's stream; pix_i = column, pix_j = row.  d_depth [H][W], d_color [H][W][3] (d_color / d_gt_color may be NULL). */
int nsk_sample_pixels(nsk_ctx* ctx, unsigned long long seed, int n, int H0, int H1, int W0, int W1, int32_t* d_pix_i, int32_t* d_pix_j);
int nsk_gather_pixels(nsk_ctx* ctx, int n, const int32_t* d_pix_i, const int32_t* d_pix_j, int H, int W, const float* d_depth,
                      const float* d_color, float* d_gt_depth, float* d_gt_color);
int nsk_rays_from_pixels(nsk_ctx* ctx, int n, const int32_t* d_pix_i, const int32_t* d_pix_j, float fx, float fy,
                         float cx, float cy, const float* d_c2w, int mode, float* d_rays_o, float* d_rays_d);
/* fused forms for a device-resident Tracker iteration (same arithmetic as the calls they replace, fewer launches):
 * nsk_rays_from_camera = nsk_camera_from_tensor + nsk_rays_from_pixels (d_c2w_out: optional 12 floats);
 * nsk_pose_step = nsk_rays_backward + nsk_camera_backward + nsk_adam_vector on the 7-vector (d_g_cam_out: optional 7 floats). */
int nsk_rays_from_camera(nsk_ctx* ctx, int n, const int32_t* d_pix_i, const int32_t* d_pix_j, float fx, float fy, float cx, float cy,
                         const float* d_cam, int mode, float* d_rays_o, float* d_rays_d, float* d_c2w_out);
int nsk_pose_step(nsk_ctx* ctx, int n, const int32_t* d_pix_i, const int32_t* d_pix_j, float fx, float fy, float cx, float cy, int mode,
                  const float* d_g_rays_o, const float* d_g_rays_d, float* d_cam, float* d_m, float* d_v, float lr, float b1, float b2,
                  float eps, int step, float* d_g_cam_out);
/* One launch for the whole ray preparation of an iteration (src/Mapper.cpp:376-427 for a window of frames, src/Tracker.cpp:44-58 for
 * one): for each of `nframes` frames, rays_per_frame times  nsk_sample_pixels (window [H0,H1) x [W0,W1), frame's own seed) ->
 * nsk_gather_pixels (frame's images, [H][W] / [H][W][3]) -> nsk_rays_from_pixels (pose = 12 floats c2w) or nsk_rays_from_camera
 * (pose_is_cam7: 7 floats quaternion + translation) -> nsk_inside_filter, with the arithmetic of those entry points (results are
 * bit-identical; tests/test_gpu_dist.py).  Outputs are [nframes * rays_per_frame] arrays, frame-major; d_keep may be NULL (no
 * filter), d_color / d_gt_color may be NULL.  The frame table is read on the host at call time (at most 64 frames per call).
 * The reference spends 3 kernels and ~15 small tensor ops per frame here; the device-resident Mapper iteration spent 16 launches. */
typedef struct nsk_frame_rays {
    const float* d_depth; const float* d_color;      /* the frame's images on the device */
    const float* d_pose;                             /* device: 12 floats (c2w rows) or 7 floats (pose vector) */
    int pose_is_cam7;
    unsigned long long seed;                         /* pixel-draw seed of this frame in this iteration */
} nsk_frame_rays;
int nsk_prepare_rays(nsk_ctx* ctx, int nframes, const nsk_frame_rays* h_frames, int rays_per_frame, int H0, int H1, int W0, int W1,
                     int H, int W, float fx, float fy, float cx, float cy, int mode, int32_t* d_pix_i, int32_t* d_pix_j,
                     float* d_gt_depth, float* d_gt_color, float* d_rays_o, float* d_rays_d, uint8_t* d_keep);
/* nsk_pose_step for every frame of a mapping window in ONE launch (bundle adjustment, src/Mapper.cpp:305-329,366-368,467-489): frame f owns
 * the rays [h_first[f], h_first[f] + h_count[f]) of the batch's pixel / ray-gradient arrays (a shard of the batch at N > 1: the part of
 * the frame's rays that falls into this rank's range, possibly none) and the pose d_cams[8 f .. 8 f + 6] with moments d_m / d_v in the same
 * layout; h_active[f] = 0 marks a frame whose pose is not optimised (the oldest frame of the window).
 *   step >= 1: gradient + Adam step of every active pose, the arithmetic of nsk_pose_step frame by frame (one GPU);
 *   step == 0: gradients only, d_g_cams[8 f + k] = d loss / d pose (zeros for inactive frames) -- the form for N > 1: register d_g_cams
 *              with nsk_grad_extra so that it is summed over the ranks with the grid gradients, then step all poses at once with
 *              nsk_adam_vector(8 * nframes, d_cams, d_g_cams, ...) (a zero gradient leaves a pose and its moments untouched).
 * d_g_cams may be NULL for step >= 1; when given it has room for 8 * nframes + 8 floats and d_g_cams[8 nframes + 1] receives the number
 * of rays of d_keep[0 .. n_keep) that take part (d_keep NULL: n_keep), the count a sharded step's ranks must add up to the batch's. */
int nsk_pose_step_multi(nsk_ctx* ctx, int nframes, const int* h_first, const int* h_count, const uint8_t* h_active, const int32_t* d_pix_i,
                        const int32_t* d_pix_j, float fx, float fy, float cx, float cy, int mode, const float* d_g_rays_o, const float* d_g_rays_d,
                        float* d_cams, float* d_m, float* d_v, float lr, float beta1, float beta2, float eps, int step, float* d_g_cams,
                        const uint8_t* d_keep, int n_keep);
/* d loss / d c2w (12 floats, overwritten) from per-ray gradients */
int nsk_rays_backward(nsk_ctx* ctx, int n, const int32_t* d_pix_i, const int32_t* d_pix_j, float fx, float fy,
                      float cx, float cy, int mode, const float* d_g_rays_o, const float* d_g_rays_d, float* d_g_c2w);
/* get_camera_from_tensor / quad2rotation (utils.h:174-210): cam = (qw,qx,qy,qz,tx,ty,tz) -> c2w [3][4] */
int nsk_camera_from_tensor(nsk_ctx* ctx, const float* d_cam, float* d_c2w);
int nsk_camera_backward(nsk_ctx* ctx, const float* d_cam, const float* d_g_c2w, float* d_g_cam);
/* inside-bbox pre-filter (src/Mapper.cpp:416-427, src/Tracker.cpp:48-58): d_keep[n] = (t >= gt_depth) */
int nsk_inside_filter(nsk_ctx* ctx, int N, const float* d_rays_o, const float* d_rays_d, const float* d_gt_depth,
                      uint8_t* d_keep);
/* Registers the NEXT batch, so that its sampling and cell sort leave the front of its own step: call it BEFORE the nsk_map_step of the current
 * batch, with the arguments the next nsk_map_step will get (same device pointers, same gt_depth_max, same flags, the next batch's ray mask
 * installed while you call it, the same render options and seed).  Nothing is launched by this call.  The nsk_map_step that follows carries
 * the registered batch's sampling in its composite launch, its backward launch carries the offsets of the cell sort and the nsk_adam_step
 * after it the placement; the next nsk_map_step finds its samples ready and starts with the forward.  Whatever has not been carried by then
 * (the call came after the step, another call needed the cell histogram in between, gt_depth_max < 0 with more than 8192 rays, a graph
 * capture) is launched at that point, as for an unprepared batch; a registered batch that is never asked for is dropped.  The rays and
 * the ground truth must stay valid and unchanged until their step has run, and must not depend on the running step's result (bundle
 * adjustment moves poses: do not prepare then).  The same holds for the ray mask: what is remembered is the BUFFER installed with
 * nsk_set_ray_mask at the time of this call, and its contents are read later -- when the batch's sampling runs inside the current step's
 * composite launch, or at the batch's own step -- so the next batch's mask must live in a different buffer from the current step's mask and
 * stay unchanged until the batch's step has run (two mask buffers used alternately, as Mapper::optimize_map does).
 * Results are those of the unprepared step.  Everything runs on the context's stream.
 * With hipGraphs: nsk_graph_begin first runs whatever of a prepared batch's sampling / sort is still pending (a capture never records it),
 * nsk_graph_launch does the same and drops a prepared batch that lives in the buffer set the graph writes (its own step samples it again). */
int nsk_map_prepare(nsk_ctx* ctx, int stage, int N, const float* d_rays_o, const float* d_rays_d, const float* d_gt_depth, float gt_depth_max,
                    unsigned flags);

/* The reference drops the rays that fail the test above before it renders them (boolean-index compaction, src/Mapper.cpp:423-427,
 * src/Tracker.cpp:55-58), which needs their count on the host.  Here they stay in place and are neutralised instead: with a mask
 * installed (d_keep [N] as written by nsk_inside_filter, device memory, must stay valid; NULL = none), nsk_map_step, nsk_track_step and
 * nsk_render_backward leave the rays with d_keep == 0 out of max(gt_depth), the Tracker's median, the loss and every gradient --
 * the same sums the reference forms over the compacted batch, with no device-to-host round trip. */
int nsk_set_ray_mask(nsk_ctx* ctx, const uint8_t* d_keep);
/* A rank that renders a SHARD of a batch (rays shard over the GPUs of a node) still needs the batch-global max(gt_depth)
 * (src/Renderer.cpp:76,93).  Every rank can hold the whole batch's ground-truth depths and keep bytes (they come from the same pixel draw:
 * nsk_prepare_rays of the full window is one small launch), so no collective is needed: with a depth-max batch installed
 * (d_gt_depth[n], d_keep[n] or NULL; n = 0 removes it) every step that is given gt_depth_max < 0 takes the maximum over THAT array
 * instead of over its own rays (inside the sampling launch, every wave for itself, while that batch has at most 8192 rays; one extra
 * single-block launch in front of the sampling beyond that).  nsk_map_prepare remembers the batch installed at registration, like the ray mask. */
int nsk_set_depth_max_batch(nsk_ctx* ctx, const float* d_gt_depth, const uint8_t* d_keep, int n);
/* plain Adam on a caller-owned vector (camera 7-vectors: src/Tracker.cpp:103, src/Mapper.cpp:305-329) */
int nsk_adam_vector(nsk_ctx* ctx, int n, float* d_p, const float* d_g, float* d_m, float* d_v, float lr, float beta1,
                    float beta2, float eps, int step);

/* ---- optimiser (torch::optim::Adam, src/Mapper.cpp:330,360-368,445-446) ---------------------------------- */
/* lr[g] for NSK_GROUP_*; groups whose gradients were produced since the last step are updated (an lr of 0
 * still advances their moments, as torch does); gradients are zeroed afterwards (optimizer.zero_grad(), :446).
 * Only masked voxels move (nsk_set_mask).  nsk_adam_reset drops the moments and step counts (the reference
 * re-creates the optimiser on every optimize_map call, :330). */
int nsk_adam_step(nsk_ctx* ctx, const float lr[NSK_NUM_GROUPS], float beta1, float beta2, float eps);
int nsk_adam_reset(nsk_ctx* ctx);

/* ---- hipGraph capture of a step ------------------------------------------------------------------------------
 * The kernels of the nsk_* calls made between nsk_graph_begin and nsk_graph_end (e.g. nsk_map_step + nsk_adam_step with fixed
 * device buffers, sizes, learning rates and a device-computed or fixed gt_depth_max) are recorded instead of run, and
 * nsk_graph_launch replays them with one launch on the context's stream; Adam's step counts advance per replay (the recorded
 * Adam node is patched with the new bias-correction constants).  Run the step once eagerly first (workspaces are sized then);
 * at most one nsk_adam_step per graph; calls that synchronise (uploads, downloads, nsk_grad_slab) are not capturable.
 * A recorded step holds buffer addresses and the optimiser masks' voxel lists as kernel arguments: after a reallocation (larger
 * batch, new grid shape) or ANY nsk_set_mask / nsk_frustum_mask call (new mask contents included) every graph is stale --
 * nsk_graph_launch then fails with a message instead of replaying; run the step eagerly once and capture again. */
int nsk_graph_begin(nsk_ctx* ctx);
int nsk_graph_end(nsk_ctx* ctx, int* graph_id);
int nsk_graph_launch(nsk_ctx* ctx, int graph_id);
int nsk_graph_destroy(nsk_ctx* ctx, int graph_id);
int nsk_zero_grads(nsk_ctx* ctx);

/* ---- multi-GPU ------------------------------------------------------------------------------------------- */
/* The gradient slab (all grid gradients, all decoder gradients, one loss scalar) is one contiguous device
 * buffer so that a mapping step needs exactly one all-reduce.  The decoder gradients of a step
 * are summed into the slab lazily (inside nsk_adam_step when nobody looks earlier): nsk_grad_slab, nsk_allreduce_grads and
 * nsk_decoder_grad_download complete that sum first, so call nsk_grad_slab after nsk_map_step, every step, before reading
 * or exchanging the slab yourself. */
int nsk_grad_slab(nsk_ctx* ctx, float** d_ptr, size_t* n_floats);
/* The exchange in compact form.  Every rank holds the same optimiser masks (nsk_set_mask / nsk_frustum_mask), and Adam discards the
 * gradient of an unmarked voxel, so only marked voxels need to travel: nsk_grad_pack gathers, into one contiguous buffer, the marked
 * voxels of the grid levels that received gradients since the last optimiser step (whole levels where no mask is installed), the
 * gradients of the trainable decoders and the 4 loss floats; after the caller's all-reduce (sum) over that buffer nsk_grad_unpack
 * writes the sums back into the slab for nsk_adam_step.  Levels the stage did not touch are not sent at all.  The marked-voxel lists
 * are rebuilt only when a mask changes (ascending voxel order, identical on every rank). */
int nsk_grad_pack(nsk_ctx* ctx, float** d_ptr, size_t* n_floats);
int nsk_grad_unpack(nsk_ctx* ctx);
/* A caller-owned device vector that travels with the exchange: nsk_grad_pack appends its n_floats (a multiple of 4, 16-byte aligned) behind
 * the loss floats, nsk_grad_unpack writes the sums back into it.  The C++ Mapper registers [8 floats per window frame: the bundle-adjustment
 * pose gradients of nsk_pose_step_multi(step = 0) | loss | kept-ray count | 6 spare] so that a sharded BA iteration still needs exactly
 * one all-reduce (SURVEY.md 8e lists the 7 (window - 1) pose floats as part of the slab).  n_floats = 0 removes it. */
int nsk_grad_extra(nsk_ctx* ctx, float* d_buf, size_t n_floats);
/* nsk_grad_pack + ncclAllReduce(sum, fp32) on the context's stream + nsk_grad_unpack; comm is an ncclComm_t (RCCL). */
int nsk_allreduce_grads(nsk_ctx* ctx, void* nccl_comm);

/* ---- introspection for benchmarks ------------------------------------------------------------------------ */
/* algorithmic bytes / flops of the last render or step call (SURVEY.md section 8d accounting) */
int nsk_last_call_stats(nsk_ctx* ctx, double* alg_bytes, double* alg_flops, int* samples);
/* per-kernel timing with HIP events recorded on the context's stream around every launch between
 * nsk_profile_begin and nsk_profile_end; _end synchronises and writes "name launches total_ms\n" lines. */
int nsk_profile_begin(nsk_ctx* ctx);
int nsk_profile_end(nsk_ctx* ctx, char* buf, size_t buf_bytes);

/* ---- test aids (tests/test_gpu_relu.py, tools/relu_flips.py; never on the product path) ------------------------- */
/* A ReLU's derivative jumps at zero, so what a gradient test can demand depends on which side of every kink the forward stood.
 * nsk_debug_relu_bits: the "input > 0" bits the forward of the last nsk_map_step / nsk_track_step / nsk_render_backward saved for decoder
 * `which` (1 middle, 2 fine, 3 colour; reference src/models/MLP.cpp:92,98 torch::relu), by SAMPLE (ray * S + s): h_bits[M][5][32] bytes.
 * nsk_debug_preact: the ReLU inputs themselves ([M][5][32] floats, device), recomputed over the same samples by the forward body of the
 * current matmul mode (N, rays = those of the last step). */
int nsk_debug_relu_bits(nsk_ctx* ctx, int which, int M, uint8_t* h_bits);
/* nsk_debug_fetch: a per-sample array of the last step's workspace, to the host: what = 0..2 the occupancy output of decoder 0..2 [M] (their sum is
 * the sigma whose relu the compositing takes, include/torchlib/utils.h:160), 3 the colour decoder's output [M][4], 4 d loss / d raw [M][4], 5 z [M]. */
int nsk_debug_fetch(nsk_ctx* ctx, int what, int M, float* h_out);
int nsk_debug_preact(nsk_ctx* ctx, int which, int N, const float* d_rays_o, const float* d_rays_d, float* d_preact);

#ifdef __cplusplus
}
#endif
#endif /* NSK_H */
