/*
 * st3r.h -- C ABI of libst3r_hip.so, the MI355X (gfx950) hot path behind the
 * starster Python API.
 *
 * The reference (phuang1024/Starst3r v0.4.0) has NO plugin/FFI boundary of its own:
 * its hot path is reached through Python calls into third-party packages
 *   starster/gs.py:76-87          gsplat.rasterization(...)            -> st3r_gs_* below
 *   starster/gs.py:126-136,153    compute_loss + loss.backward()       -> st3r_loss_l1_ssim, *_bwd
 *   starster/gs.py:37,159-161     6x torch.optim.Adam.step             -> st3r_adam_step
 *   starster/reconstruct.py:371-406  optimize_loop (alignment)         -> st3r_align_*
 *   starster/reconstruct.py:97    forward_mast3r -> fast_reciprocal_NNs-> st3r_recip_nn
 * so this header DEFINES the boundary a maintainer would bind with ctypes (see
 * INTEGRATION.md).  Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - every pointer argument is a DEVICE pointer owned by the caller unless the name
 *     ends in _host; the library never frees or retains caller memory past the call;
 *   - scratch lives in a grow-only arena inside the ctx;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it unless
 *     documented otherwise;
 *   - every function returns 0 on success or a negative ST3R_ERR_* code; the message is
 *     available from st3r_last_error() (thread local).  No C++ exception crosses the ABI;
 *   - a ctx belongs to one (process, GPU) and is not thread safe.
 *
 * Layouts (floats unless noted)
 *   means [N,3]  quats [N,4] (w,x,y,z)  scales [N,3] RAW  opacities [N] RAW
 *   sh    [N,sh_stride]  with the degree<=1 coefficients in the first 12 floats of each
 *         row ([4,3]); for the reference's shN tensor [N,24,3] sh_stride = 72
 *   viewmats [C,4,4] world->camera row major, Ks [C,3,3], campos [C,3] = inverse(viewmats)[:, :3, 3]
 *   splats [C*N,12]: x y opacity | conic a b c | r g b | depth | radius (int32 bits) | 0
 *          pair id pid = cam*N + gaussian ("dense" ids; radius == 0 <=> culled)
 *   v_splats [C*N,12]: v_x v_y v_opacity | v_conic a b c | v_r v_g v_b | 0 0 0
 *   grads / adam m / adam v [23*N]: blocks means[3N] quats[4N] scales[3N] opacities[N] sh4[12N]
 *   images [C,H,W,3], alpha [C,H,W], last_ids int32 [C,H,W]
 */
#ifndef ST3R_H
#define ST3R_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST3R_VERSION 100 /* 0.1.0 */

#define ST3R_OK 0
#define ST3R_ERR_INVALID (-1)  /* bad argument */
#define ST3R_ERR_HIP (-2)      /* a HIP runtime call failed */
#define ST3R_ERR_CAPACITY (-3) /* caller-supplied capacity too small */
#define ST3R_ERR_NOMEM (-4)
#define ST3R_ERR_PEER (-5)     /* the previous exchanged training step failed on a rank of the communicator: nobody applied it */

#define ST3R_SPLAT_STRIDE 12
#define ST3R_MAX_VIEWS 256    /* views per call (the camera table lives in LDS: 128 B per view) */
#define ST3R_GRAD_PER_GAUSSIAN 23

typedef struct st3r_ctx st3r_ctx;

int st3r_version(void);
const char* st3r_last_error(void);
int st3r_ctx_create(int device, st3r_ctx** out);
int st3r_ctx_destroy(st3r_ctx* ctx);
/* bytes currently held by the ctx arena (diagnostics) */
int64_t st3r_ctx_arena_bytes(st3r_ctx* ctx);

/* Test hook: copy `bytes` of a scratch buffer left by the last fused step (st3r_gs_train_fwd_bwd /
 * st3r_gs_render) into caller memory `dst` (device).  which: 0 sorted pair ids int32 [n_isects],
 * 1 tile offsets int32 [C*tiles], 2 splat records float [C*N*12], 3 inclusive tile scan int32 [C*N]. */
int st3r_ctx_peek(st3r_ctx* ctx, void* stream, int which, void* dst, int64_t bytes);

/* Per-stage timing with HIP events recorded on the caller's stream around every stage of the
 * fused steps (no host synchronisation while enabled).  st3r_ctx_get_stage_ms synchronises
 * the device, writes the accumulated milliseconds and sample counts of the ST3R_NUM_STAGES
 * stages and resets them.  Stage order: project, scan, emit, sort, offsets, blend_fwd, loss,
 * blend_bwd, project_bwd, adam, sort_depth (names from st3r_stage_name).
 * enable: 0 = off, 1 = every stage, 2 + s = stage s only (two events per step instead of twenty-two). */
#define ST3R_NUM_STAGES 11
int st3r_ctx_set_profiling(st3r_ctx* ctx, int enable);
/* test hook. bit 0: the blend forward walks every staged record in every wave (no per-quadrant relevance test):
 * images must come out bit-identical, which is how the culling is validated at full size.
 * st3r_ctx_peek(which = 8 / 9): rgb [C,H,W,3] / alpha [C,H,W] of the last st3r_gs_train_fwd_bwd call; which = 10: the 16
 * int32 control words of the fused steps ([8..10] = bias, sentinel key and pass count of the segmented level-1 sort).
 * bit 2 (4): the training calls' level-1 sort on (camera | depth) keys in four passes instead of per-camera segments.
 * (the other bits: csrc/common.h, st3r_ctx::debug_flags) */
int st3r_ctx_set_debug(st3r_ctx* ctx, int flags);
/* Waits for the record count of the last asynchronous training step (st3r_gs_train_fwd_bwd / st3r_gs_train_step with
 * stats_host == NULL) and reports it like the next training call would: ST3R_ERR_CAPACITY if that step outgrew its
 * buffers (its Adam update was then skipped on the device: repeat the step).  For the end of a training loop
 * (starster/gs.py:143-166 returns after the last iteration); a no-op when nothing is in flight. */
int st3r_ctx_settle(st3r_ctx* ctx);
/* Gives the ctx's scratch back to the device allocator (the arena is grow-only otherwise: a 1 M-Gaussian / 8 x 1080p step
 * holds ~6 GB, one rank of configs[4] ~150 GB): waits for the device, settles like st3r_ctx_settle (whose code it returns),
 * frees every scratch buffer.  The ctx stays valid -- the next call allocates what it needs again, one synchronising
 * step; communicator, settings and sizing hints are kept.  For a host that alternates phases on one GPU (Mast3r
 * inference <-> training), or hands the GPU to another process. */
int st3r_ctx_release_scratch(st3r_ctx* ctx);
int st3r_ctx_get_stage_ms(st3r_ctx* ctx, double* ms_out, int64_t* counts_out);
const char* st3r_stage_name(int stage);

/* ------------------------------------------------------------------------------------
 * Path C -- 3DGS rasterization stages (replace gsplat.rasterization, starster/gs.py:76-87)
 * ---------------------------------------------------------------------------------- */

/* projection + degree-1 SH colour for every (camera, gaussian).  gsplat
 * fully_fused_projection + spherical_harmonics + clamp_min(c+0.5,0).
 * Also accumulates sum(sigmoid(opacities)), sum(exp(scales)) (the regularisers of
 * starster/gs.py:132-134) and the number of visible pairs into reg_sums[3]
 * (double, device, NOT zeroed here, may be NULL). */
int st3r_gs_project_sh(st3r_ctx* ctx, void* stream, int N, int C, const float* means, const float* quats,
                       const float* scales, const float* opacities, const float* sh, int sh_stride,
                       const float* viewmats, const float* Ks, const float* campos, int width, int height,
                       int tile_size, float eps2d, float near_plane, float far_plane, float radius_clip,
                       float* splats, int32_t* tiles_per_gauss, double* reg_sums);

/* inclusive scan of tiles_per_gauss -> cum_tiles; total copied to *n_isects_host
 * (this call synchronises the stream). gsplat isect_tiles pass 1 + torch.cumsum. */
int st3r_gs_isect_scan(st3r_ctx* ctx, void* stream, int64_t n_pairs, const int32_t* tiles_per_gauss,
                       int32_t* cum_tiles, int64_t* n_isects_host);

/* gsplat isect_tiles pass 2: key = cam << (32+tile_bits) | tile << 32 | depth bits,
 * value = dense pair id. */
int st3r_gs_isect_emit(st3r_ctx* ctx, void* stream, int N, int C, const float* splats,
                       const int32_t* cum_tiles, int tile_size, int tile_w, int tile_h, int64_t n_isects,
                       int64_t* isect_ids, int32_t* flatten_ids);

/* stable ascending sort of (key,value) on bits [0,end_bit).  cub DeviceRadixSort::SortPairs
 * as used by gsplat.  Inputs may be clobbered. */
int st3r_gs_sort(st3r_ctx* ctx, void* stream, int64_t n_isects, int end_bit, int64_t* isect_ids,
                 int32_t* flatten_ids, int64_t* isect_ids_sorted, int32_t* flatten_ids_sorted);

/* The sort underneath st3r_gs_sort, for any key shape of the pipeline: stable ascending LSD radix sort of
 * (key, int32 value) pairs on key bits [begin_bit, end_bit), key_bytes = 4 or 8 (unsigned).  Hand-written
 * onesweep (one histogram launch + one chained-scan launch per 8-bit digit); inputs are left untouched,
 * vals_in/vals_out may be NULL (keys only).  Scratch comes from the ctx. */
int st3r_radix_sort_pairs(st3r_ctx* ctx, void* stream, int key_bytes, int64_t n, int begin_bit, int end_bit,
                          const void* keys_in, const int32_t* vals_in, void* keys_out, int32_t* vals_out);

/* gsplat isect_offset_encode: offsets [C,tile_h,tile_w] */
int st3r_gs_offsets(st3r_ctx* ctx, void* stream, int64_t n_isects, const int64_t* isect_ids_sorted, int C,
                    int tile_w, int tile_h, int32_t* offsets);

/* gsplat rasterize_to_pixels forward (no background, RGB) */
int st3r_gs_blend_fwd(st3r_ctx* ctx, void* stream, int C, int width, int height, int tile_size, int tile_w,
                      int tile_h, const float* splats, const int32_t* offsets, const int32_t* flatten_ids,
                      int64_t n_isects, float* rgb, float* alpha, int32_t* last_ids);

/* gsplat rasterize_to_pixels backward.  v_alpha may be NULL (== 0).  Must follow
 * st3r_gs_blend_fwd of the same arguments on the same ctx (the forward leaves per-wave
 * contribution bitmasks in ctx scratch).  Gradients are accumulated without HBM atomics:
 * each (record, tile) pair owns a slot in ctx scratch, and v_splats[pid] (n_pairs = C*N
 * records, fully written) is the in-order sum of the pair's slots.  cum_tiles is the
 * inclusive scan from st3r_gs_isect_scan. */
int st3r_gs_blend_bwd(st3r_ctx* ctx, void* stream, int C, int width, int height, int tile_size, int tile_w,
                      int tile_h, const float* splats, const int32_t* offsets, const int32_t* flatten_ids,
                      int64_t n_isects, const float* alpha, const int32_t* last_ids, const float* v_rgb,
                      const float* v_alpha, const int32_t* cum_tiles, int64_t n_pairs, float* v_splats);

/* backward of st3r_gs_project_sh, summed over cameras, plus the regulariser gradients
 *   reg_views * opac_fac * d mean|sigmoid(o)|   and   reg_views * scale_fac * d mean|exp(s)|
 * (starster/gs.py:132-134, added once per view :150-152).  Writes grads [23*N]. */
int st3r_gs_project_sh_bwd(st3r_ctx* ctx, void* stream, int N, int C, const float* means, const float* quats,
                           const float* scales, const float* opacities, const float* sh, int sh_stride,
                           const float* viewmats, const float* Ks, const float* campos, int width, int height,
                           float eps2d, const float* splats, const float* v_splats, float reg_views,
                           float opac_fac, float scale_fac, float* grads);

/* L1 + SSIM of C views (starster/gs.py:126-130; torchmetrics SSIM data_range=1).
 *   loss_c = w_l1 * mean|gt - r| + w_ssim * (1 - SSIM(gt, r))
 * sums [C,2] (double, device): per view  sum|gt-r|  and  sum of the interior SSIM map
 * (zeroed by this call).  v_render [C,H,W,3] = d(sum_c loss_c)/d render; may be NULL. */
int st3r_loss_l1_ssim(st3r_ctx* ctx, void* stream, int C, int height, int width, const float* render,
                      const float* gt, float w_l1, float w_ssim, double* sums, float* v_render);

/* Two of SSIM's five windowed moments -- conv(gt) and conv(gt^2) -- depend on the ground truth alone, and the ground
 * truth is the same in every iteration of a training call (starster/gs.py:149-152 passes scene.imgs; torchmetrics
 * recomputes them every call, starster/gs.py:129).  st3r_loss_gt_moments writes them once: moments [C,H,W,3,2] =
 * (conv(gt), conv(gt^2)) per pixel and channel, zeros outside the interior (H-10) x (W-10).
 * st3r_ctx_set_gt_moments registers such a buffer for the images at `gt` [C,H,W,3]: from then on st3r_loss_l1_ssim,
 * st3r_gs_train_fwd_bwd / _step and st3r_gs_raster_train read the moments instead of convolving gt whenever their
 * ground-truth pointer is `gt` or a whole-view offset into it (view shards, view chunks) with the same height and
 * width -- bit-identical sums and gradients (same taps in the same order).  Both buffers stay the CALLER's and must
 * stay unchanged while registered; gt = NULL or moments = NULL clears the registration. */
int st3r_loss_gt_moments(st3r_ctx* ctx, void* stream, int C, int height, int width, const float* gt, float* moments);
int st3r_ctx_set_gt_moments(st3r_ctx* ctx, const float* gt, const float* moments, int C, int height, int width);

/* fused Adam over the 23 active scalars per gaussian (replaces the 6 torch.optim.Adam of
 * starster/gs.py:37,159-161; sh0 and SH rows 4..23 never receive gradient and are
 * left untouched, their Adam update is exactly 0).  `step` is 1-based. */
int st3r_adam_step(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                   float* opacities, float* sh, int sh_stride, const float* grads, float* m, float* v, double lr,
                   double beta1, double beta2, double eps, int step);

/* The same update restricted to the scalars [i0, i1) of the 23N-float gradient / moment buffers (buffer order: means[3N]
 * quats[4N] scales[3N] opacities[N] sh4[12N]) -- the piece a rank owns after a reduce-scatter of the gradients
 * (st3r_gs_train_step with ST3R_EXCHANGE=rs_ag does exactly this inside).  param_stage (may be NULL): a 23N-float buffer
 * in buffer order that receives the new parameter values of the piece, the payload of the parameter all-gather.
 * st3r_params_from_stage then copies the scalars OUTSIDE [i0, i1) from such a (gathered) buffer into the parameter
 * tensors, for indices < limit (the tail past limit is each rank's own: see comm.hip). */
int st3r_adam_step_range(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                         float* opacities, float* sh, int sh_stride, const float* grads, float* m, float* v,
                         double lr, double beta1, double beta2, double eps, int step, int64_t i0, int64_t i1,
                         float* param_stage);
int st3r_params_from_stage(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                           float* opacities, float* sh, int sh_stride, const float* param_stage, int64_t i0,
                           int64_t i1, int64_t limit);

/* ------------------------------------------------------------------------------------
 * Fused train step, first half: everything of one iteration of starster/gs.py:143-153
 * (render all C local views -> loss -> backward) using ctx scratch only.
 *   grads   [23*N]  written (not accumulated)
 *   loss_out        device float; receives this rank's loss (sum over its views, regularisers
 *                   added reg_views times)
 *   stats_host[4]   optional host int64: n_visible_pairs, n_isects actually sorted/blended (after exact tile
 *                   culling), arena bytes, n_isects of the reference algorithm (gsplat's 3-sigma squares)
 * Host synchronisation.  The number of tile intersections is produced on the device.  With stats_host != NULL
 * the call copies it back and synchronises the stream once (exact statistics, exactly sized scratch).  With
 * stats_host == NULL and a count known from an earlier call on the same (N, C, width, height) the call is fully
 * asynchronous: scratch is sized from the previous count (+25 %), every kernel reads the count from device
 * memory, and the count is checked when the NEXT call into this ctx starts -- if a step ever outgrows its
 * capacity (its surplus records are dropped, never written out of bounds) that next call returns
 * ST3R_ERR_CAPACITY and the one after it falls back to the synchronous path.
 * More than 2^31 tile intersections (the counts are int32; 5 M Gaussians x 8 views at 4K get there): the call walks
 * its views in chunks -- each a complete rasterize -> loss -> backward whose parameter gradients add up, the loss
 * being a sum over views -- doubling the chunk count until every chunk fits; the count sticks to the ctx and chunked
 * calls always take the synchronous path.  Found on the asynchronous path, the overflow makes the NEXT call return
 * ST3R_ERR_CAPACITY (that step's gradients were incomplete: repeat it) and later calls are chunked.
 * The second half is an (optional) all-reduce of `grads` by the caller, then st3r_adam_step.
 * ---------------------------------------------------------------------------------- */
int st3r_gs_train_fwd_bwd(st3r_ctx* ctx, void* stream, int N, int C, const float* means, const float* quats,
                          const float* scales, const float* opacities, const float* sh, int sh_stride,
                          const float* viewmats, const float* Ks, const float* campos, const float* gt_images,
                          int width, int height, float ssim_fac, float opac_fac, float scale_fac, float* grads,
                          float* loss_out, int64_t* stats_host);

/* Inference render of C views into caller buffers using ctx scratch (starster/gs.py:47-88
 * without `info`). */
int st3r_gs_render(st3r_ctx* ctx, void* stream, int N, int C, const float* means, const float* quats,
                   const float* scales, const float* opacities, const float* sh, int sh_stride,
                   const float* viewmats, const float* Ks, const float* campos, int width, int height,
                   float* rgb, float* alpha, int64_t* stats_host);

/* ------------------------------------------------------------------------------------
 * Path B -- global alignment.  Runs the whole two-stage optimisation of
 * starster/reconstruct.py:116-457 (`sparse_scene_optimizer_slam`: stage 1 = loss_3d for niter1
 * steps at lr1, stage 2 = loss_2d for niter2 steps at lr2; Adam betas (0.9,0.9), cosine schedule,
 * quaternion renormalisation, loss_dust3r with weight dust_weight) on the device, two launches per
 * iteration, no host synchronisation.  All arrays are device pointers.
 *   views:    imsizes [C,2] (W,H as float), base_focals [C], median [C] (median of the raw core
 *             depths), core [C,G] (core depths / median), min_focals/max_focals [C] (focal clamp)
 *   anchors:  anchor_pix [A,2], anchor_idx int32 [A] (index into the view's core depths),
 *             anchor_off [A], anchor_img int32 [A]
 *   rows:     corr_* = loss_3d rows (anchor ids a1,a2, weight = conf/sum conf);
 *             c2d_*  = loss_2d rows (pixel in img1, anchor id of the 3-D point, img1, weight);
 *             dust_* = regression-fallback rows (anchor id, target point in img2's frame, img2, weight)
 *   chain:    root, edges int32 [C-1,2] (parent, child) in root-outward order
 *   params (in/out, the reference's optim_params): pps [C,2] NORMALISED by the image size,
 *             log_focals [C], quats [C,4] (x,y,z,w), trans [C,3], log_sizes [C]
 *   work:     >= 66*C + 8 floats of scratch
 *   outputs:  cam_out [C,24] = R[9] (cam2w rotation, row major) T[3] f cx cy A B base_focal ...,
 *             with depthmap = A + B*core;  pts_out [A,3] world points of the anchors (may be NULL);
 *             losses_out [niter1+niter2]
 * ---------------------------------------------------------------------------------- */
int st3r_align_run(st3r_ctx* ctx, void* stream, int C, int G, int n_anchors, const float* imsizes,
                   const float* base_focals, const float* median, const float* core, const float* min_focals,
                   const float* max_focals, const float* anchor_pix, const int32_t* anchor_idx,
                   const float* anchor_off, const int32_t* anchor_img, int n_corr, const int32_t* corr_a1,
                   const int32_t* corr_a2, const float* corr_w, int n_c2d, const float* c2d_pix,
                   const int32_t* c2d_a2, const int32_t* c2d_img1, const float* c2d_w, int n_dust,
                   const int32_t* dust_a1, const float* dust_tgt, const int32_t* dust_img2, const float* dust_w,
                   int root, int n_edges, const int32_t* edges, float lr1, int niter1, float lr2, int niter2,
                   float dust_weight, float* pps, float* log_focals, float* quats, float* trans, float* log_sizes,
                   float* work, int64_t work_floats, float* cam_out, float* pts_out, float* losses_out);

/* The same optimisation with the non-default options of sparse_scene_optimizer_slam that only change constants of the
 * loop (starster/reconstruct.py:118-122; st3r_align_run = the reference's own configuration :61-69):
 *   lr_host    HOST array of niter1 + niter2 learning rates, one per iteration = the caller's
 *              schedule(iter / niter, lr_base, 0) (reconstruct.py:384-386), or NULL for cosine_schedule;
 *   gamma1/2/d exponents of the robust losses gamma_loss(g) of loss_3d (loss1), loss_2d (loss2) and loss_dust3r
 *              (lossd): rho(d) = (d + off)^g - off^g with off = (1/g)^(1/(g-1)); g = 1 is the plain distance;
 *   opt_pp     0: the principal points stay fixed in the second stage (reconstruct.py:436).
 *   opt_depth  (reconstruct.py:437: the core depths are parameters of the second stage too) is on when depth_csr_off is
 *              not NULL.  The caller groups the rows of stage 2 -- the n_c2d loss_2d rows, then the n_dust regression
 *              rows, numbered in that order -- by the core depth (img * G + anchor_idx) of the row's anchor (c2d_a2 /
 *              dust_a1): depth_csr_off int32 [C*G + 1], depth_csr_rows int32 [n_c2d + n_dust].  A core depth's
 *              gradient is the sum over its rows in that order (no float atomics).  `core` is then updated IN PLACE;
 *              depth_work: >= n_c2d + n_dust + 3*C*G floats of scratch whose last C*G floats return the core depths
 *              the exported results belong to (the reference's results are one optimiser step behind its parameters,
 *              reconstruct.py:379-380, 405-406): depthmap = A + B * that.
 * shared_intrinsics, exp_depth, lora_depth and per-image `init` freezes are not implemented. */
int st3r_align_run_opts(st3r_ctx* ctx, void* stream, int C, int G, int n_anchors, const float* imsizes,
                        const float* base_focals, const float* median, float* core, const float* min_focals,
                        const float* max_focals, const float* anchor_pix, const int32_t* anchor_idx,
                        const float* anchor_off, const int32_t* anchor_img, int n_corr, const int32_t* corr_a1,
                        const int32_t* corr_a2, const float* corr_w, int n_c2d, const float* c2d_pix,
                        const int32_t* c2d_a2, const int32_t* c2d_img1, const float* c2d_w, int n_dust,
                        const int32_t* dust_a1, const float* dust_tgt, const int32_t* dust_img2, const float* dust_w,
                        int root, int n_edges, const int32_t* edges, float lr1, int niter1, float lr2, int niter2,
                        float dust_weight, float* pps, float* log_focals, float* quats, float* trans, float* log_sizes,
                        float* work, int64_t work_floats, float* cam_out, float* pts_out, float* losses_out,
                        const float* lr_host, float gamma1, float gamma2, float gammad, int opt_pp,
                        const int32_t* depth_csr_off, const int32_t* depth_csr_rows, float* depth_work,
                        int64_t depth_work_floats);

/* ------------------------------------------------------------------------------------
 * Path A -- matching.  The nearest-neighbour query of Mast3r's fast_reciprocal_NNs with
 * dist='dot' (reached from starster/reconstruct.py:97):
 *     nn_out[q] = argmax_j  queries[q] . db[j]     (smallest j on ties), score_out[q] = that maximum
 * queries [n,dim], db [m,dim] row major float32 (dim must be 24, Mast3r's descriptor size),
 * nn_out int32 [n], score_out float [n] or NULL.  fp32 MFMA block-matmul with fused arg-max.
 * st3r_recip_nn below runs the reciprocal iteration around it on the device.
 * ---------------------------------------------------------------------------------- */
int st3r_nn_dot_argmax(st3r_ctx* ctx, void* stream, const float* queries, int n, const float* db, int m, int dim,
                       int32_t* nn_out, float* score_out);

/* The whole reciprocal iteration of fast_reciprocal_NNs(A, B, subsample_or_initxy1=subsample, dist='dot')
 * (starster/reconstruct.py:97 passes subsample=8) resident on the device, no host synchronisation:
 *   seeds = flat indices x + W1*y on the grid np.mgrid[S//2:H1:S, S//2:W1:S]   (n = st3r_recip_nn_seed_count)
 *   repeat max_iter (10) times, only for seeds that have not converged:
 *       idx2 = argmax_B(A[idx1] . B^T); converged if idx2 did not change
 *       idx1 = argmax_A(B[idx2] . A^T); converged if idx1 did not change
 * descA [H1*W1, dim], descB [H2*W2, dim] float32, dim = 24.  Outputs int32 [n]: idx1_out, idx2_out, and
 * notyet_out (0 = converged: a reciprocal match; the caller keeps those, then de-duplicates and sorts --
 * merge_corres -- which stays host-side bookkeeping). */
int st3r_recip_nn_seed_count(int H1, int W1, int subsample);
int st3r_recip_nn(st3r_ctx* ctx, void* stream, const float* descA, int H1, int W1, const float* descB, int H2, int W2,
                  int dim, int subsample, int max_iter, int32_t* idx1_out, int32_t* idx2_out, int32_t* notyet_out);

/* ----------------------------------------------------------------------------------
 * C7 -- refinement hooks of gsplat.MCMCStrategy() with its default hyper-parameters, as the reference
 * drives them from starster/gs.py:43-45 (construction), :146-147 (pre-backward no-op) and :163-164
 * (step_post_backward(step, info, lr=1e-3)) when run_3dgs_optim(enable_pruning=True).
 * All three interpret `opacities` as logits and `scales` as logs (the renderer reads them raw, SURVEY B-1).
 * Random draws are Philox4x32-10(key = seed, counter = (index, stream, step)): replicas that hold the same
 * parameters make identical decisions.  Buffers are caller-owned device memory, updated in place.
 *
 * st3r_mcmc_relocate: Gaussians with sigmoid(opacity) <= min_opacity take the parameters of alive ones
 *   drawn with probability ~ sigmoid(opacity); drawn sources get the relocation opacity/scale; the Adam
 *   moments (block layout [23N], or NULL) of the sources are zeroed.  n_dead_host (or NULL) receives the
 *   number of relocated Gaussians (forces one stream sync).
 * st3r_mcmc_add: the arrays have room for N + n_new rows; rows [N, N + n_new) are filled the same way
 *   (n_new = min(cap_max, int(1.05 N)) - N is the caller's business, as is extending the Adam state by zeros).
 * st3r_mcmc_noise: means += Sigma(quats, exp(scales)) @ (randn * sigmoid_100(1 - sigmoid(opacity) - 0.995) * scaler)
 * st3r_ctx_peek(which = 4..7) reads back the integer sampling state of the last relocate/add call:
 *   4 = uint64 inclusive prefix sums of the 24-bit weights [N], 5 = uint32 dead mask [N],
 *   6 = int32 sampled source per row [n] followed by int32 destination row [n] (relocate), 7 = uint32 draw counts [N].
 * ---------------------------------------------------------------------------------- */
int st3r_mcmc_relocate(st3r_ctx* ctx, void* stream, int N, float* means, float* quats, float* scales,
                       float* opacities, float* sh0, float* shN, int shN_floats, float* adam_m, float* adam_v,
                       float min_opacity, uint64_t seed, uint32_t step, int64_t* n_dead_host);
int st3r_mcmc_add(st3r_ctx* ctx, void* stream, int N, int n_new, float* means, float* quats, float* scales,
                  float* opacities, float* sh0, float* shN, int shN_floats, float min_opacity, uint64_t seed,
                  uint32_t step);
int st3r_mcmc_noise(st3r_ctx* ctx, void* stream, int N, float* means, const float* quats, const float* scales,
                    const float* opacities, float scaler, uint64_t seed, uint32_t step);
/* The same for the rows [row_offset, row_offset + n) of the Gaussian set (the pointers address the first of them):
 * the draws are keyed by the global row, so the shards of a Gaussian-sharded job perturb exactly like the whole set. */
int st3r_mcmc_noise_rows(st3r_ctx* ctx, void* stream, int n, int64_t row_offset, float* means, const float* quats,
                         const float* scales, const float* opacities, float scaler, uint64_t seed, uint32_t step);

/* ----------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, views sharded, Gaussians and Adam state replicated; the one
 * exchange step is a sum all-reduce of the [23N] gradient buffer per iteration over RCCL.  The reference is
 * single-process (starster/gs.py:143-164); sharding is valid because its loss is a sum over views
 * (gs.py:149-152).  RCCL is bound at run time (dlopen: the library named by ST3R_RCCL_LIB if set, else the librccl.so.1
 * already in the process, else a fresh one); without it only these entry points fail.
 *
 *   st3r_comm_unique_id  rank 0 creates the 128-byte id, the host distributes it by its own means
 *   st3r_comm_init       every rank joins (collective; the ctx then owns the communicator)
 *   st3r_comm_attach     alternatively adopt a caller-owned ncclComm_t (never destroyed by the library)
 *   st3r_grad_allreduce  in place, asynchronous on `stream`; a no-op for a single replica
 *   st3r_gs_train_step   st3r_gs_train_fwd_bwd -> st3r_grad_allreduce -> st3r_adam_step in one call:
 *                        one iteration of starster/gs.py:143-164 for this rank's views.  loss_out holds
 *                        this rank's part of the loss (sum over ranks = the reference's loss).
 *
 * The exchange inside st3r_gs_train_step takes one of four forms, a setting of the ctx (st3r_comm_set_exchange):
 *   ST3R_EXCHANGE_ALLREDUCE  one all-reduce of the 23N floats after the backward, then Adam (default)
 *   ST3R_EXCHANGE_RANGES     the projection backward runs per Gaussian range, a range's all-reduce overlaps the next
 *                            range's backward and the previous range's Adam
 *   ST3R_EXCHANGE_RS_AG      reduce-scatter -> Adam on the rank's piece -> all-gather of the parameters; the Adam
 *                            moments are then maintained on the own piece only (st3r_comm_allgather_pieces replicates
 *                            them again before the form is left)
 *   ST3R_EXCHANGE_DIRECT     the same three steps without RCCL on the data path: every rank exports its gradient buffer,
 *                            a parameter staging buffer and a row of flags through HIP IPC (set up collectively over the
 *                            communicator the first time the form is used, and again when the Gaussian set outgrows
 *                            the buffers); per step a rank READS its 1/w piece of the gradients from all peers over
 *                            the point-to-point links and sums it in rank order (one owner per element: replicas are
 *                            bit-identical by construction), runs Adam on the piece, and READS the other pieces of
 *                            the updated parameters from their owners; two device-side barriers per step over the
 *                            exported flags (bounded: a peer that never arrives is a failed step, not a hang), the
 *                            step's status word travels with BOTH of them.  A barrier that times out is FATAL for the
 *                            communicator: the rank that saw it skips what follows the barrier, every later training
 *                            call on it returns ST3R_ERR_PEER without exchanging anything (the peers then time out in
 *                            their next barrier and end the same way), and the replicas may differ by that one step --
 *                            tear the communicator down and restore the parameters.  The flags live in uncached
 *                            device memory exported over HIP IPC; where that cannot be had the form is refused
 *                            (ST3R_ERR_HIP from the first step).  EXPERIMENTAL: exercised between processes that share
 *                            one device only (tests/test_gpu_multi.py), never yet across two devices.  Moments as under
 *                            RS_AG.  st3r_comm_destroy is COLLECTIVE once this form has been used (nobody may unmap a
 *                            buffer a peer still reads; the wait for the peers is bounded by ST3R_XBAR_TIMEOUT_MS --
 *                            past it the window is leaked instead of unmapped).
 * A communicator starts with the form the environment variable ST3R_EXCHANGE names (allreduce | ranges | rs_ag | direct; read once,
 * when the communicator is created or attached; never written by the library), else with the all-reduce.  All ranks must
 * use the same form.
 *
 * Errors under a communicator: a step whose forward / backward fails on one rank (that rank returns the error) is applied
 * on NO rank -- the failing rank still takes part in every collective, a max-reduced status word guards the update on the
 * device -- and the next training call (or st3r_ctx_settle) of EVERY rank, the failing one included, returns ST3R_ERR_PEER
 * without issuing a collective: all ranks skip that call together and then repeat the step (a host that retries on one
 * rank only desynchronises the collective sequence).  Replicas stay identical, nobody hangs.  The guarantee covers the
 * step's computation; a HIP / RCCL call of the exchange itself that fails (after the status word has travelled) is
 * returned by that rank once it has issued the step's remaining collectives -- the other ranks cannot be told, the
 * communicator should be torn down.  The status word guards the update kernels of st3r_gs_train_step only; the
 * stand-alone st3r_adam_step / st3r_adam_step_range / st3r_params_from_stage never look at it.
 * ---------------------------------------------------------------------------------- */
#define ST3R_COMM_ID_BYTES 128
#define ST3R_EXCHANGE_ALLREDUCE 0
#define ST3R_EXCHANGE_RANGES 1
#define ST3R_EXCHANGE_RS_AG 2
#define ST3R_EXCHANGE_DIRECT 3
int st3r_comm_set_exchange(st3r_ctx* ctx, int form);
int st3r_comm_get_exchange(st3r_ctx* ctx, int* form);
int st3r_comm_allgather_pieces(st3r_ctx* ctx, void* stream, float* buf, int64_t count);
int st3r_comm_unique_id(char* id_out);
int st3r_comm_init(st3r_ctx* ctx, int world_size, int rank, const char* id);
int st3r_comm_attach(st3r_ctx* ctx, void* rccl_comm, int world_size, int rank);
int st3r_comm_destroy(st3r_ctx* ctx);
int st3r_comm_world(st3r_ctx* ctx, int* world_size, int* rank);
int st3r_grad_allreduce(st3r_ctx* ctx, void* stream, float* grads, int64_t count);
int st3r_gs_train_step(st3r_ctx* ctx, void* stream, int N, int C, float* means, float* quats, float* scales,
                       float* opacities, float* sh, int sh_stride, const float* viewmats, const float* Ks,
                       const float* campos, const float* gt_images, int width, int height, float ssim_fac,
                       float opac_fac, float scale_fac, float* grads, float* m, float* v, double lr, double beta1,
                       double beta2, double eps, int step, float* loss_out, int64_t* stats_host);

/* ----------------------------------------------------------------------------------
 * Dense point extraction between alignment and 3DGS seeding: what
 * starster/scene.py:148 takes from `scene.get_dense_pts3d(clean_depth=True)` (Mast3r SparseGA [U]).
 * Views are concatenated; view v owns the dense pixels [view_start[v], view_start[v+1]) in raster order.
 *   cam          [C,24]  the per-view rows st3r_align_run writes (R(9) T(3) f cx cy A B ...), i.e. the optimised
 *                        cam2w / intrinsics and depthmap_v = A + B * core_depth[v]
 *   core_depth   [C,G]   normalised core depths (the alignment's input/output), idxs index into a view's row
 *   pixels [n,2], idxs int32 [n], offsets [n]: every dense pixel as an anchor of its view's core depthmap
 * st3r_dense_unproject -> pts_out [n,3] world points, zcam_out [n] depth in the own camera.
 * st3r_dense_clean (dust3r clean_pointcloud [U], tol = 0.001, bad_conf = 0 upstream): conf [n] is lowered in
 *   place, view by view in the reference's order, for points that project inside another view, lie in front of
 *   that view's depth by more than tol (relative) and are less confident than the pixel they land on.
 *   sizes_hw int32 [C,2] = (H, W) of every view; max_view_pixels = largest H*W.
 * ---------------------------------------------------------------------------------- */
int st3r_dense_unproject(st3r_ctx* ctx, void* stream, int C, int G, int n, const int32_t* view_start,
                         const float* pixels, const int32_t* idxs, const float* offsets, const float* core_depth,
                         const float* cam, const float* base_focals, float* pts_out, float* zcam_out);
int st3r_dense_clean(st3r_ctx* ctx, void* stream, int C, int max_view_pixels, const int32_t* view_start,
                     const int32_t* sizes_hw, const float* cam, const float* pts, const float* zcam, float tol,
                     float bad_conf, float* conf);

/* ----------------------------------------------------------------------------------
 * Canonical-data condensation: the arithmetic of Mast3r's prepare_canonical_data between the
 * matching and the alignment, called from starster/reconstruct.py:101 (mast3r/cloud_opt/sparse_ga.py [U], not
 * vendored by the reference).  All pointers are device pointers, float32.
 * st3r_canon_view (canonical_view, mode 'avg-angle'): ptmaps [n,H,W,3] / confs [n,H,W] = the n predictions of an
 *   image in its own frame -> canon [H,W,3] confidence-weighted mean (weights conf - 0.999), cconf [H,W] =
 *   sum w^2 / sum w, canon2 [H,W] = depth of every pixel relative to its subsample-block centre from averaged
 *   elevation angles.  H, W multiples of subsample.
 * st3r_focal_weiszfeld (dust3r estimate_focal_knowing_depth, 'weiszfeld'): focal_out [1] on the device;
 *   min_focal / max_focal are multiples of the 60-degree-FOV focal max(H,W) / (2 tan 30deg) (upstream 0.5 / 3.5).
 * st3r_anchor_offsets (anchor_depth_offsets): xy [n,2] pixels of an image's correspondences -> idx_out int32 [n]
 *   (index of the block's core depth, row-major over the (H/s, W/s) grid), off_out [n] = canon2(pixel) / canon2(block
 *   centre).
 * ---------------------------------------------------------------------------------- */
int st3r_canon_view(st3r_ctx* ctx, void* stream, int n, int H, int W, int subsample, const float* ptmaps,
                    const float* confs, float* canon, float* canon2, float* cconf);
int st3r_focal_weiszfeld(st3r_ctx* ctx, void* stream, int H, int W, const float* canon, float ppx, float ppy,
                         float min_focal, float max_focal, float* focal_out);
/* the same for n_views images of one size in ONE launch: canon [n_views,H,W,3] -> focal_out [n_views] */
int st3r_focal_weiszfeld_batch(st3r_ctx* ctx, void* stream, int n_views, int H, int W, const float* canon, float ppx,
                               float ppy, float min_focal, float max_focal, float* focal_out);
int st3r_anchor_offsets(st3r_ctx* ctx, void* stream, int64_t n, int H, int W, int subsample, const float* canon2,
                        const float* xy, int32_t* idx_out, float* off_out);

/* ----------------------------------------------------------------------------------
 * Gaussian-sharded multi-GPU mode (an alternative to the gradient all-reduce when every rank owns one or two
 * views): rank r owns the Gaussians [r N/w, (r+1) N/w) -- parameters and Adam state are NOT replicated -- and
 * the views [r C, (r+1) C).  Per iteration:
 *   1. st3r_gs_project_sh on the own Gaussians for ALL views          -> records [V, N/w, 12]
 *   2. all-to-all: every rank receives the records of all Gaussians for its views -> [C, N, 12]
 *   3. st3r_gs_raster_train(N, C, records, gt, ...)                   -> v_records [C, N, 12], image loss
 *   4. all-to-all back: owners receive [V, N/w, 12]
 *   5. st3r_gs_project_sh_bwd on the own Gaussians (pass opac_fac, scale_fac scaled by (N/w)/N so that the
 *      regularisers stay means over all N) and st3r_adam_step on the shard.
 * Exchanged per rank and iteration: 2 x 48 B x C x N (w-1)/w, against 2 x 92 B x N (w-1)/w for the gradient
 * all-reduce; no replicated optimiser work.  starst3r_amd/dist.py holds the two exchanges.
 * ---------------------------------------------------------------------------------- */
int st3r_gs_raster_train(st3r_ctx* ctx, void* stream, int N, int C, const float* records, const float* gt_images,
                         int width, int height, float ssim_fac, float* v_records, float* loss_out,
                         int64_t* stats_host);

#ifdef __cplusplus
}
#endif
#endif /* ST3R_H */
